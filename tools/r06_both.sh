#!/bin/bash
# round 6: the 16-bit MLP backward's halves as one launch -- parity, then the A/B at the driver's invocation
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_ah}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tiles.py tests/test_gpu_step.py -q -x -k "fused_encode or mlp_backward or step_over_the_list or train_step_matches or default_precision" -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/${T}_tests.txt
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:8]))"; }
{ for e in 0 1 0 1 0 1; do echo "== cfg2 driver invocation, mlp_bwd_one_launch=$e"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --mlp-bwd-one-launch $e; done; } 2>&1 | tee gpurun_out/${T}_both.txt
