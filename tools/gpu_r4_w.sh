#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_fixture.py -x -q -m gpu -k "fused_raymarch or wave_ray_marcher or reference_driven or sample_points" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 > gpurun_out/r04_final5_bench_driver_invocation_nocpu.json
python -c "
import json
d=json.loads(open('gpurun_out/r04_final5_bench_driver_invocation_nocpu.json').read())
print('headline', round(d['ms_per_step'],4), 'spread', d['step_ms_spread'], 'settled', d['ms_per_step_settled'], 'round', d['round_ms_per_step'], 'dense', d['ms_per_step_dense_backward'], 'captured', d['captured_step_ms_per_step'], 'loss', d['loss'], 'flags', d['flags'], d['kernel_ms_warmup'])"
