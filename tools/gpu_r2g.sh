#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/r02_e_bench.log | tail -1 > gpurun_out/r02_e_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r02_e_bench.json')); print(d['ms_per_step'], d['captured_step_ms_per_step'], d['roofline']['avg_ms'], d['kernel_ms_warmup'])"
