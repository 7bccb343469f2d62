#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== default: one level's gathers in flight, 60 reps each"; timeout 600 python tools/fused_debug.py 60 2>&1 | grep "quarter" | tee -a gpurun_out/${1}_fused_debug.txt
timeout 600 python -m pytest tests/test_gpu_ops.py::test_fused_encode_mlp_forward_equals_the_two_launches tests/test_gpu_step.py tests/test_gpu_reference_fixture.py tests/test_gpu_tiles.py -q --timeout=300 -p no:cacheprovider 2>&1 | tail -4
ARGS="--no-cpu-baseline --no-extra-configs --steps 200 --warmup 20 --round-steps 0 --settle 0"
run() { echo -n "[$1 | $2] "; env $1 python bench.py $ARGS $2 2>gpurun_out/${T}_err.txt | tail -1 | tee -a gpurun_out/${T}_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print(round(d['ms_per_step'],4), 'dense', round(d['ms_per_step_dense_backward'],4), 'p50', round(d['step_ms_spread']['p50'],4) if d.get('step_ms_spread') else None, {n: k[n] for n in list(k)[:4]})" || tail -5 gpurun_out/${T}_err.txt; }
T=$1
run "X=0" ""
run "X=0" "--mlp reference"
run "X=0" "--precision fp16"
run "X=0" "--unfused"
