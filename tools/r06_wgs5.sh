#!/bin/bash
# round 6: persistent scatter workgroups per CU inside the merged launch at cfg5
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_ar}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0 --steps 40 --warmup 60 --keyframes 8"
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:6]))"; }
{ for w in 0 3 6 8 12 0; do echo "== cfg5 --scatter-wgs $w"; run $CFG5 --scatter-wgs $w; done; } 2>&1 | tee gpurun_out/${T}_wgs5.txt
