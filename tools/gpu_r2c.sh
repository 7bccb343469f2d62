#!/bin/bash
# round 2, third GPU pass: kernel trace of the cfg5 workload (wide network) + wide tests after the dW remap
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -q -p no:cacheprovider -k "wide" 2>&1 | tail -5 > gpurun_out/r2c_tests.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof5
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof5 -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 > $R/gpurun_out/r2c_prof.log 2>&1
cd $R; python tools/prof_summary.py gpurun_out/prof5/bench_results.db 2>&1 | head -30 | cut -c1-170 > gpurun_out/r2c_cfg5_kernel_stats.txt
rm -rf gpurun_out/prof5
cat gpurun_out/r2c_tests.log; cat gpurun_out/r2c_cfg5_kernel_stats.txt; tail -2 gpurun_out/r2c_prof.log | cut -c1-300
