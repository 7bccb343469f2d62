#!/bin/bash
# one gpurun call: parity tests, smoke, bench, rocprof kernel trace of the bench.  Every leg has its own timeout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_mesh.py tests/test_gpu_fullsize.py tests/test_gpu_rays.py tests/test_gpu_dp.py -m gpu -q > gpurun_out/tests.log 2>&1
tail -4 gpurun_out/tests.log | cut -c1-300
if [ "$RUNNER" = "1" ]; then timeout 400 python -m pytest tests/test_gpu_runner.py -m gpu -q -x -s > gpurun_out/tests_runner.log 2>&1; grep -E "runner losses|pose translation|Chamfer|passed|failed|^E  " gpurun_out/tests_runner.log | head -12 | cut -c1-400; fi
timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 420 python bench.py ${BENCH_ARGS:---steps 100 --warmup 10 --no-cpu-baseline} > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -1 gpurun_out/bench.log | cut -c1-3400; tail -6 gpurun_out/bench.err
export TMPDIR=/tmp
rm -rf gpurun_out/prof
cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; python tools/prof_summary.py gpurun_out/prof/bench_results.db 2>&1 | head -14 | cut -c1-150
if [ "$PMC" = "1" ]; then KF=16 bash tools/gpu_pmc.sh; fi
if [ "$DIST" = "1" ]; then
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dist.log 2> gpurun_out/bench_dist.err
  tail -1 gpurun_out/bench_dist.log | cut -c1-600; tail -3 gpurun_out/bench_dist.err
fi
