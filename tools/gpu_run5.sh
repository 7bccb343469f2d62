#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03_f}
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_ops.py tests/test_gpu_step.py -x -q -k "wide or tiles" 2>&1 | tail -6 > gpurun_out/${T}_tests.txt; tail -4 gpurun_out/${T}_tests.txt
timeout 900 python bench.py --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --steps 40 --warmup 100 --keyframes 16 --no-cpu-baseline 2>gpurun_out/${T}_bench_cfg5.log | tail -1 > gpurun_out/${T}_bench_cfg5.json; cut -c1-260 gpurun_out/${T}_bench_cfg5.json; tail -3 gpurun_out/${T}_bench_cfg5.log
cd /tmp
rm -rf $R/gpurun_out/prof_5; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_5 -o bench -- python $R/bench.py --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --steps 20 --warmup 100 --keyframes 8 --no-cpu-baseline > $R/gpurun_out/${T}_prof_bench5.json 2>$R/gpurun_out/${T}_prof5.log
cd $R
python tools/step_timeline.py $(find gpurun_out/prof_5 -name "*_results.db" | head -1) 105 > gpurun_out/${T}_cfg5_timeline.txt 2>&1; cat gpurun_out/${T}_cfg5_timeline.txt
find gpurun_out -name "*.db" -size +30M -delete
