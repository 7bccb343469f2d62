#!/bin/bash
# bench.py under a list of environment settings on ONE box (A/B of step-orchestration knobs):
#   gpurun -- 'bash tools/gpu_ab_env.sh "A=1" "B=1" "A=1 B=1"'      (the plain run is always first and last)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ARGS=${BENCH_ARGS:-"--no-cpu-baseline --steps 400 --warmup 50"}
run() { echo -n "[$1] "; env $1 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'dense', round(d.get('ms_per_step_dense_backward') or 0,4), 'dom', round(d['roofline']['avg_ms'],4))"; }
run "X=0"
for e in "$@"; do run "$e"; done
run "X=0"
