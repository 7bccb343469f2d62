#!/bin/bash
# round 6: the one-chain backward tail against the two-stream tail AT THE DRIVER'S INVOCATION (64 keyframes, 20 steps after 5)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_y}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_step.py -q -x -k "one_stream or hash_backward_over_the_list or graphed_step or dyn_step" -p no:cacheprovider 2>&1 | tail -3
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'p50dev', d.get('ms_per_step_p50_timed'), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'two-branch', round(d.get('captured_step_two_branches_ms_per_step') or 0,4), 'loss', d['loss'])"; }
{ for e in 0 1 0 1 0 1; do echo "== driver invocation one_stream=$e"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --round-steps 0 --one-stream $e; done; } 2>&1 | tee gpurun_out/${T}_one_stream_driver.txt
