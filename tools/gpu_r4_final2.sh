#!/bin/bash
# round 4, last code state: the whole GPU suite + the driver's bench invocation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_final2}
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --timeout=600 -p no:cacheprovider 2>&1 | tail -22 > gpurun_out/${T}_gpu_tests.txt; tail -5 gpurun_out/${T}_gpu_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_driver.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-200 gpurun_out/${T}_bench_driver_invocation.json
