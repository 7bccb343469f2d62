#!/bin/bash
# round 4, call A: the whole GPU suite + the bench lines of the two network shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_a}
timeout 1700 python -m pytest tests -m gpu -q --durations=12 --timeout=600 -p no:cacheprovider 2>&1 | tail -70 > gpurun_out/${T}_gpu_tests.txt; tail -50 gpurun_out/${T}_gpu_tests.txt
for a in "" "--mlp reference"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 $a 2>gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench_$(echo $a | tr -d ' -').json
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_$(echo $a | tr -d ' -').json').read())
print('$a', d['ms_per_step'], d.get('ms_per_step_dense_backward'), d.get('ms_per_step_first_steps'), {k: round(v,4) for k,v in d.get('kernel_ms_warmup',{}).items()})
PY
done
