#!/bin/bash
# round 4, call F: the whole GPU suite on the final default code + bench.py as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_f}
timeout 1700 python -m pytest tests -m gpu -q --durations=12 --timeout=600 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${T}_gpu_tests.txt; tail -30 gpurun_out/${T}_gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_driver.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-400 gpurun_out/${T}_bench_driver_invocation.json; tail -3 gpurun_out/${T}_bench_driver.log
