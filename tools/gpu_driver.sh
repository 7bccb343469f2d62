#!/bin/bash
# What the driver does at round end, in one gpurun call:  gpurun -- 'bash tools/gpu_driver.sh <tag> [pytest args]'
#   build() (rebuilds any library whose stamp lags its sources) -> pytest -m gpu -x -q -> smoke() -> bench.py defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r05}; shift
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.log 2>&1 || { tail -20 gpurun_out/${T}_build.log; exit 1; }
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 --timeout=600 -p no:cacheprovider "$@" 2>&1 | tail -70 > gpurun_out/${T}_gpu_tests.txt; tail -40 gpurun_out/${T}_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/${T}_smoke.txt
if [ -z "$NO_BENCH" ]; then
  # (the driver's own command line: BENCH_r0N.json records `python3 bench.py --gpus 1 --steps 20 --warmup 5`)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${T}_bench.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-400 gpurun_out/${T}_bench_driver_invocation.json
fi
