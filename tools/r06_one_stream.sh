#!/bin/bash
# round 6: the backward tail as one chain (scatter + dL/dx in one launch, pose kernels behind the LDS level) vs two streams
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_x}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 600 python -m pytest tests/test_gpu_tiles.py -q -x -k "hash_backward_over_the_list" -p no:cacheprovider 2>&1 | tail -3
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0 --steps 40 --warmup 60 --keyframes 8"
CFG2="--no-cpu-baseline --no-extra-configs --steps 100 --warmup 50 --round-steps 0 --settle 300 --keyframes 16"
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'dense', round(d.get('ms_per_step_dense_backward') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'loss', d['loss'], 'flags', d['flags'], '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:6]))"; }
{ for e in 0 1 0 1; do echo "== cfg2 one_stream=$e"; run $CFG2 --one-stream $e; done
  for e in 0 1; do echo "== cfg5 one_stream=$e"; run $CFG5 --one-stream $e; done; } 2>&1 | tee gpurun_out/${T}_one_stream.txt
