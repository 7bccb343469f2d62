#!/bin/bash
# round 4, call B: the fused forward's tests, then A/B of the forward forms and of the backward workgroup shapes on ONE box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_b}
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_render.py tests/test_gpu_step.py tests/test_gpu_tiles.py tests/test_gpu_reference_fixture.py "tests/test_gpu_fullsize.py::test_fullsize_step_matches_oracle[cfg2-fp16x3]" -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/${T}_tests.txt; tail -12 gpurun_out/${T}_tests.txt
ARGS="--no-cpu-baseline --no-extra-configs --steps 200 --warmup 20 --round-steps 0"
run() { echo -n "[$1 | $2] "; env $1 python bench.py $ARGS $2 2>gpurun_out/${T}_err.txt | tail -1 | tee -a gpurun_out/${T}_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print(round(d['ms_per_step'],4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'dense', round(d['ms_per_step_dense_backward'],4), 'p50', d['step_ms_spread']['p50'] if d.get('step_ms_spread') else None, {n: k[n] for n in list(k)[:7]})" || tail -5 gpurun_out/${T}_err.txt; }
run "X=0" ""
run "X=0" "--unfused"
run "NOF_LIB=$PWD/bundlesdf_amd/ab_enc8g4.so" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_enc12g4.so" ""
run "NOF_LIB=$PWD/bundlesdf_amd/ab_bwd8.so" ""
run "X=0" "--mlp reference"
run "X=0" "--mlp reference --unfused"
run "X=0" ""
