#!/bin/bash
# bench.py with the regular library and with A/B builds of it on ONE box:  gpurun -- 'bash tools/gpu_ab_lib.sh <tag> ab_x.so ab_y.so'
# (the regular library first and last; BENCH_ARGS overrides the bench options)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
T=$1; shift
ARGS=${BENCH_ARGS:-"--no-cpu-baseline --no-extra-configs --steps 200 --warmup 50 --round-steps 0 --settle 300"}
run() { echo -n "[$1] "; if [ "$1" = regular ]; then python bench.py $ARGS 2>/dev/null; else NOF_LIB=$PWD/bundlesdf_amd/$1 python bench.py $ARGS 2>/dev/null; fi | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'dense', round(d.get('ms_per_step_dense_backward') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:9]))"; }
{ run regular; for so in "$@"; do run $so; done; run regular; } | tee gpurun_out/${T}.txt
