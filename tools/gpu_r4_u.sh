#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wave_ray_marcher" 2>&1 | tail -4
for v in 0 1 1 0; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --trace-kernel $v 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('trace_kernel', $v, 'headline', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'settled', round(d['ms_per_step_settled'],4), 'round', round(d['round_ms_per_step'],4), 'raymarch', d['kernel_ms_warmup'].get('nof_raymarch_sample'), 'loss', round(d['loss'],6), 'flags', d['flags'])"
done
