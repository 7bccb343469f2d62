#!/bin/bash
tag=${1:-r04_n}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tiles.py tests/test_gpu_step.py tests/test_gpu_reference_fixture.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_driver$i.json 2> gpurun_out/${tag}_bench_driver$i.log
python - gpurun_out/${tag}_bench_driver$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('headline', round(d['ms_per_step'],4), 'spread', d.get('step_ms_spread'), 'settled', d.get('ms_per_step_settled'), 'round', d.get('round_ms_per_step'), 'dense', d.get('ms_per_step_dense_backward'), 'captured', d.get('captured_step_ms_per_step'))
print(d.get('kernel_ms_warmup'))
PY
done
