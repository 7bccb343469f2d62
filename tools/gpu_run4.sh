#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_ops.py tests/test_gpu_step.py -x -q 2>&1 | tail -8 > gpurun_out/r03_e_tests.txt; tail -5 gpurun_out/r03_e_tests.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>gpurun_out/r03_e_bench.log | tail -1 > gpurun_out/r03_e_bench.json; cut -c1-260 gpurun_out/r03_e_bench.json; tail -3 gpurun_out/r03_e_bench.log
timeout 900 python bench.py --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --steps 40 --warmup 100 --keyframes 16 --no-cpu-baseline 2>gpurun_out/r03_e_bench_cfg5.log | tail -1 > gpurun_out/r03_e_bench_cfg5.json; cut -c1-260 gpurun_out/r03_e_bench_cfg5.json; tail -3 gpurun_out/r03_e_bench_cfg5.log
cd /tmp
rm -rf $R/gpurun_out/prof_e; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e -o bench -- python $R/bench.py --steps 60 --warmup 200 --no-cpu-baseline --keyframes 16 > $R/gpurun_out/r03_e_prof_bench.json 2>$R/gpurun_out/r03_e_prof.log
rm -rf $R/gpurun_out/prof_e5; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e5 -o bench -- python $R/bench.py --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --steps 20 --warmup 100 --keyframes 8 --no-cpu-baseline > $R/gpurun_out/r03_e_prof_bench5.json 2>$R/gpurun_out/r03_e_prof5.log
cd $R
python tools/step_timeline.py $(find gpurun_out/prof_e -name "*_results.db" | head -1) 240 > gpurun_out/r03_e_timeline.txt 2>&1; cat gpurun_out/r03_e_timeline.txt
python tools/prof_summary.py $(find gpurun_out/prof_e5 -name "*_results.db" | head -1) > gpurun_out/r03_e_cfg5_kernel_stats.txt 2>&1; head -22 gpurun_out/r03_e_cfg5_kernel_stats.txt | cut -c1-60,73-112
python tools/step_timeline.py $(find gpurun_out/prof_e5 -name "*_results.db" | head -1) 105 > gpurun_out/r03_e_cfg5_timeline.txt 2>&1; cat gpurun_out/r03_e_cfg5_timeline.txt
find gpurun_out -name "*.db" -size +30M -delete
