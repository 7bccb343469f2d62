#!/bin/bash
tag=${1:-r04_o}
mkdir -p gpurun_out
for v in 0 1 0 1; do
HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_kernarg$v.json 2> gpurun_out/${tag}_bench_kernarg$v.log
python - gpurun_out/${tag}_bench_kernarg$v.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('HIP_FORCE_DEV_KERNARG', sys.argv[2], 'headline', round(d['ms_per_step'],4), 'spread', d.get('step_ms_spread'), 'settled', d.get('ms_per_step_settled'), 'round', d.get('round_ms_per_step'), 'dense', d.get('ms_per_step_dense_backward'), 'captured', d.get('captured_step_ms_per_step'))
PY
done
env | grep -i "^HIP\|^HSA\|^ROC\|^GPU_\|^AMD" | head -20
