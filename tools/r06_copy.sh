#!/bin/bash
# round 6: the weight-image copy with four pieces per thread in flight -- parity of every MLP kernel, the fixed cost of the merged backward, the bench
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_at}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_erratum.py tests/test_gpu_ops.py tests/test_gpu_tiles.py tests/test_gpu_step.py -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/${T}_tests.txt
python tools/mlp_bwd_fixed_probe.py 2>&1 | tail -3 | tee gpurun_out/${T}_mlp_bwd_fixed.txt
run() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50dev', round(d.get('ms_per_step_p50_timed') or 0,4), 'settled', round(d.get('ms_per_step_settled') or 0,4), 'captured', round(d.get('captured_step_ms_per_step') or 0,4), 'round', round(d.get('round_ms_per_step') or 0,4), 'fwd', round(d['roofline']['hash_fwd']['avg_ms'],4), 'dom', round(d['roofline']['avg_ms'],4))"; }
{ for i in 1 2 3; do echo "== driver invocation"; run --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs; done; } 2>&1 | tee gpurun_out/${T}_bench.txt
