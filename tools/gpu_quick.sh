#!/bin/bash
# a few GPU tests with hard per-test timeouts:  gpurun -- 'bash tools/gpu_quick.sh "<pytest selection>"'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }   # never measure a stale library
timeout ${2:-400} python -m pytest $1 -q -x --timeout=150 -p no:cacheprovider --durations=5 2>&1 | tail -25
