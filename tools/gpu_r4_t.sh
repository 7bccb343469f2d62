#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tiles.py tests/test_gpu_reference_fixture.py -x -q -m gpu 2>&1 | tail -2
for v in 0 0 4; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --scatter-wgs $v 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('scatter_wgs', $v, 'headline', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'settled', round(d['ms_per_step_settled'],4), 'round', round(d['round_ms_per_step'],4), 'dense', round(d['ms_per_step_dense_backward'],4))"
done
