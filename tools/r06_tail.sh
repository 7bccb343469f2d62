#!/bin/bash
# round 6: the optimiser launch with the pose sums and the next step's prologue inside (nof_adam_step_tail) against the three calls
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_z}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_step.py -q -x -k "one_stream or adam_step_tail or graphed_step or dyn_step or overflow or graph_capturable or marches" -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/${T}_tests.txt
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), 'loss', d['loss'])"; }
{ for e in 0 1 0 1 0 1; do echo "== driver invocation fused_tail=$e"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --fused-tail $e; done; } 2>&1 | tee gpurun_out/${T}_tail_driver.txt
