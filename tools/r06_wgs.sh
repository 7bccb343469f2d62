#!/bin/bash
# round 6: persistent scatter workgroups per CU inside the merged { scatter | dL/dx } launch, at the driver's invocation
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_aq}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), 'dom', round(d['roofline']['avg_ms'],4))"; }
{ for w in 0 2 3 6 8 0 2 3 6 8; do echo "== --scatter-wgs $w"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --scatter-wgs $w; done; } 2>&1 | tee gpurun_out/${T}_wgs.txt
