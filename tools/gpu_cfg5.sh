#!/bin/bash
# cfg5 (wide network) check: parity tests of the wide path, then the bench with a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -q -k "wide" 2>&1 | tail -1
rm -rf /tmp/prof5; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 > /tmp/prof5.json 2>/dev/null)
python -c "import json; d=json.loads(open('/tmp/prof5.json').read().strip().splitlines()[-1]); print('cfg5 ms/step', d['ms_per_step'])"
python tools/prof_summary.py $(find /tmp/prof5 -name "*_results.db" | head -1) > gpurun_out/cfg5_kernel_stats.txt; grep "k_wide\|k_hash\|k_adam\|k_reduce" gpurun_out/cfg5_kernel_stats.txt | cut -c1-60,73-110
