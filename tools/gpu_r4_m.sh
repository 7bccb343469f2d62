#!/bin/bash
tag=${1:-r04_m}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_reference_fixture.py tests/test_gpu_rays.py tests/test_gpu_render.py tests/test_gpu_chain.py tests/test_gpu_runner.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
for v in 0 1 0 1; do
NOF_REDUCE_SIDE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_side$v.json 2> gpurun_out/${tag}_bench_side$v.log
python - gpurun_out/${tag}_bench_side$v.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('reduce_on_side', sys.argv[2], 'headline', round(d['ms_per_step'],4), 'spread', d.get('step_ms_spread'), 'settled', d.get('ms_per_step_settled'), 'round', d.get('round_ms_per_step'), 'dense', d.get('ms_per_step_dense_backward'))
print(d.get('kernel_ms_warmup'))
PY
done
