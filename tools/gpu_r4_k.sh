#!/bin/bash
tag=${1:-r04_k}
R=$(pwd); mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_tiles.py tests/test_gpu_render.py tests/test_gpu_reference_fixture.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_driver.json 2> gpurun_out/${tag}_bench_driver.log
python - gpurun_out/${tag}_bench_driver.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('headline', round(d['ms_per_step'],4), 'spread', d.get('step_ms_spread'), 'settled', d.get('ms_per_step_settled'), 'round', d.get('round_ms_per_step'), 'dense', d.get('ms_per_step_dense_backward'))
print(d.get('kernel_ms_warmup'))
PY
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_l
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace -d $R/gpurun_out/prof_l -o b -- python $R/bench.py --steps 30 --warmup 60 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes 16 > $R/gpurun_out/${tag}_lag_bench.json 2>$R/gpurun_out/${tag}_lag.log
db=$(find $R/gpurun_out/prof_l -name "*.db" | head -1)
ls -la $db
python $R/tools/launch_lag.py $db 70 > $R/gpurun_out/${tag}_launch_lag.txt 2>&1
python $R/tools/launch_lag.py $db 71 >> $R/gpurun_out/${tag}_launch_lag.txt 2>&1
rm -rf $R/gpurun_out/prof_l
head -60 $R/gpurun_out/${tag}_launch_lag.txt | cut -c1-150
