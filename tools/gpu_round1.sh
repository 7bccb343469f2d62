#!/bin/bash
# one gpurun call: parity tests, smoke, bench, rocprof kernel trace of the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -q > gpurun_out/tests.log 2>&1
tail -15 gpurun_out/tests.log | cut -c1-300
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.log | cut -c1-3000; tail -5 gpurun_out/bench.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; ls gpurun_out/prof | head; find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -r head -25
