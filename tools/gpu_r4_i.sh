#!/bin/bash
# deferred table Adam: tests that touch the step, host enqueue time, the driver's invocation with and without it
tag=${1:-r04_i}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_step.py tests/test_gpu_render.py tests/test_gpu_tiles.py tests/test_gpu_chain.py -x -q -m gpu -s 2>&1 | grep -v "^rays \|^translation\|^sc_factor" | tail -45 > gpurun_out/${tag}_tests.txt
tail -3 gpurun_out/${tag}_tests.txt
for d in 1 0; do
  NOF_DEFER_ADAM=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_defer$d.json 2> gpurun_out/${tag}_bench_defer$d.log
  python - gpurun_out/${tag}_bench_defer$d.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(sys.argv[1], 'headline', round(d['ms_per_step'],4), 'spread', d.get('step_ms_spread'), 'settled', d.get('ms_per_step_settled'), 'round', d.get('round_ms_per_step'), 'dense', d.get('ms_per_step_dense_backward'), 'captured', d.get('captured_step_ms_per_step'))
PY
done
NOF_DEFER_ADAM=1 python tools/host_probe.py 2>&1 | grep -v "^rays \|^translation\|^sc_factor" | head -30 > gpurun_out/${tag}_host_probe_defer1.txt
NOF_DEFER_ADAM=0 python tools/host_probe.py 2>&1 | grep "400 steps" > gpurun_out/${tag}_host_probe_defer0.txt
cat gpurun_out/${tag}_host_probe_defer1.txt | head -8; cat gpurun_out/${tag}_host_probe_defer0.txt
