#!/bin/bash
# round 3, first GPU contact of the work-list backward: new parity tests, then a bench line + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_reference_fixture.py -x -q 2>&1 | tail -25 > gpurun_out/r03_a_tests.txt; tail -12 gpurun_out/r03_a_tests.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>gpurun_out/r03_a_bench.log | tail -1 > gpurun_out/r03_a_bench.json; cut -c1-400 gpurun_out/r03_a_bench.json; tail -5 gpurun_out/r03_a_bench.log
cd /tmp
rm -rf $R/gpurun_out/prof_a; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_a -o bench -- python $R/bench.py --steps 60 --warmup 200 --no-cpu-baseline --keyframes 16 > $R/gpurun_out/r03_a_prof_bench.json 2>$R/gpurun_out/r03_a_prof.log
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_a -name "*_results.db" | head -1) > gpurun_out/r03_a_kernel_stats.txt 2>&1; head -24 gpurun_out/r03_a_kernel_stats.txt | cut -c1-60,73-112
find gpurun_out -name "*.db" -size +30M -delete
