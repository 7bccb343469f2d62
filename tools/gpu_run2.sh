#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu --deselect tests/test_gpu_chain.py --deselect tests/test_gpu_fullsize.py::test_fullsize_step_matches_oracle -x -q 2>&1 | tail -25 > gpurun_out/r03_c_tests.txt; tail -8 gpurun_out/r03_c_tests.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>gpurun_out/r03_c_bench.log | tail -1 > gpurun_out/r03_c_bench.json; cut -c1-300 gpurun_out/r03_c_bench.json; tail -4 gpurun_out/r03_c_bench.log
cd /tmp
rm -rf $R/gpurun_out/prof_c; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py --steps 60 --warmup 200 --no-cpu-baseline --keyframes 16 > $R/gpurun_out/r03_c_prof_cench.json 2>$R/gpurun_out/r03_c_prof.log
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_c -name "*_results.db" | head -1) > gpurun_out/r03_c_kernel_stats.txt 2>&1; head -24 gpurun_out/r03_c_kernel_stats.txt | cut -c1-60,73-112
python tools/step_timeline.py $(find gpurun_out/prof_c -name "*_results.db" | head -1) 240 > gpurun_out/r03_c_timeline.txt 2>&1; cat gpurun_out/r03_c_timeline.txt
find gpurun_out -name "*.db" -size +30M -delete
