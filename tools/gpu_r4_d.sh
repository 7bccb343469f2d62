#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_d}
timeout 300 python tools/fused_debug.py 2>&1 | grep -v "^sc_factor\|^transl\|^rays " | tee gpurun_out/${T}_fused_debug.txt
ARGS="--no-cpu-baseline --no-extra-configs --steps 200 --warmup 20 --round-steps 0 --settle 0"
run() { echo -n "[$1 | $2] "; env $1 python bench.py $ARGS $2 2>gpurun_out/${T}_err.txt | tail -1 | tee -a gpurun_out/${T}_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print(round(d['ms_per_step'],4), 'dense', round(d['ms_per_step_dense_backward'],4), 'p50', round(d['step_ms_spread']['p50'],4) if d.get('step_ms_spread') else None, {n: k[n] for n in list(k)[:4]})" || tail -5 gpurun_out/${T}_err.txt; }
run "X=0" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_skew2.so" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_skew4.so" ""
run "X=0" ""
timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_dp.py -q --timeout=400 -p no:cacheprovider 2>&1 | tail -8
