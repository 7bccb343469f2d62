#!/bin/bash
# persistent workgroups per CU of the table scatter in the driver's (early) regime
mkdir -p gpurun_out
for v in 3 4 3 4 3 4 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --scatter-wgs $v 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('scatter_wgs', $v, 'headline', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'settled', round(d['ms_per_step_settled'],4), 'round', round(d['round_ms_per_step'],4), 'dense', round(d['ms_per_step_dense_backward'],4))"
done > gpurun_out/r04_s_scatter_wgs.txt
cat gpurun_out/r04_s_scatter_wgs.txt
