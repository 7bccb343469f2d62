#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_fixture.py -x -q -m gpu -k "trace or wave or sample_points or fixture or reference" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 > gpurun_out/r04_final4_bench_driver_invocation_nocpu.json
python -c "
import json
d=json.loads(open('gpurun_out/r04_final4_bench_driver_invocation_nocpu.json').read())
print('headline', round(d['ms_per_step'],4), d['value'], 'spread', d['step_ms_spread'], 'settled', d['ms_per_step_settled'], 'round', d['round_ms_per_step'], 'dense', d['ms_per_step_dense_backward'], 'captured', d['captured_step_ms_per_step'], 'roofline', d['roofline']['frac'], d['kernel_ms_warmup'])"
