#!/bin/bash
# round 6: what the bench's own timing events cost the timed region (5 per step in rounds 3-5) -- the driver's invocation with the
# events of the region at stride 1 / 1 (as before), 4 / 4 (the default) and none
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_an}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'max', round(d.get('ms_step_max_timed') or 0,4), 'spread p50', round(d['step_ms_spread']['p50'],4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), 'roof', round(d['roofline']['frac'],4), d['roofline']['avg_ms'], 'fwd', d['roofline'].get('hash_fwd',{}).get('avg_ms'), d['timing_events']['events_per_timed_step'])"; }
{ for e in "1 1" "4 4" "1000 1000" "1 1" "4 4" "1000 1000"; do set -- $e; echo "== driver invocation, --event-stride $1 --marker-stride $2"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --event-stride $1 --marker-stride $2; done; } 2>&1 | tee gpurun_out/${T}_events.txt
