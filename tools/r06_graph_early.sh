#!/bin/bash
# captured step against eager in the regime the driver's 20 steps run in (no settling in between)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_ad}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
run() { python bench.py "$@" 2>&1 | grep "timed region done\|captured\|settled\|dense" | cut -c1-150; }
{ for i in 1 2 3; do echo "== --settle 0"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --round-steps 0 --settle 0; done
  echo "== --settle 300 --steps 100"; run --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extra-configs --round-steps 0 --settle 300; } 2>&1 | tee gpurun_out/${T}_graph_early.txt
