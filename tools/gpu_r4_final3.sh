#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_final3}
timeout 300 python -m pytest tests/test_gpu_step.py "tests/test_gpu_fullsize.py::test_fullsize_step_matches_oracle[cfg2-fp16x3]" tests/test_gpu_tiles.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/${T}_tests.txt; cat gpurun_out/${T}_tests.txt
timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_driver.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-200 gpurun_out/${T}_bench_driver_invocation.json
