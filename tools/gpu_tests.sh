#!/bin/bash
# the whole GPU suite, as the driver runs it (no -x: every failure is listed)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }   # never measure a stale library
timeout 1500 python -m pytest tests -m gpu -q --durations=15 --timeout=400 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${1:-r03}_gpu_tests.txt; tail -45 gpurun_out/${1:-r03}_gpu_tests.txt
