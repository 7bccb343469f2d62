#!/bin/bash
# round 6: the tile scan shared by several workgroups -- tests, then cfg2 at the driver's invocation and cfg5
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_af}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_tiles.py -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/${T}_tests.txt
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0 --steps 40 --warmup 60 --keyframes 8"
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:8]))"; }
{ for i in 1 2 3; do echo "== cfg2 driver invocation"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs; done
  for i in 1 2; do echo "== cfg5"; run $CFG5; done; } 2>&1 | tee gpurun_out/${T}_scan.txt
