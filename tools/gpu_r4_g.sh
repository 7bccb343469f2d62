#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_g}
ARGS="--no-cpu-baseline --no-extra-configs --steps 200 --warmup 20 --round-steps 0"
run() { echo -n "[$1 | $2] "; env $1 python bench.py $ARGS $2 2>gpurun_out/${T}_err.txt | tail -1 | tee -a gpurun_out/${T}_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print(round(d['ms_per_step'],4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'dense', round(d['ms_per_step_dense_backward'],4), 'p50', round(d['step_ms_spread']['p50'],4), {n: k[n] for n in list(k)[:4]})" || tail -5 gpurun_out/${T}_err.txt; }
run "X=0" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_rolled.so" ""
run "X=0" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_rolled.so" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_rolled.so" "--mlp reference"
run "X=0" "--mlp reference"
NOF_LIB=$PWD/bundlesdf_amd/ab_rolled.so timeout 300 python -m pytest tests/test_gpu_ops.py::test_fused_encode_mlp_forward_equals_the_two_launches tests/test_gpu_ops.py::test_fused_forward_is_repeatable -q --timeout=300 -p no:cacheprovider 2>&1 | tail -3
