#!/bin/bash
# round 4, final: the whole GPU suite, the printed parity figures, the driver's bench invocation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_h}
timeout 1700 python -m pytest tests -m gpu -q --durations=12 --timeout=600 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/${T}_gpu_tests.txt; tail -25 gpurun_out/${T}_gpu_tests.txt
bash tools/gpu_numbers.sh $T
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_driver.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-330 gpurun_out/${T}_bench_driver_invocation.json
