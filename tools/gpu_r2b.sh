#!/bin/bash
# round 2, second GPU pass: wide-network kernels + the tests fixed after the first pass
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_runner.py -m gpu -q -p no:cacheprovider -s -k "wide or default_precision or mlp_backward or loss_scale or checkpoint" 2>&1 | grep -v "^z_in_out\|^ERROR sample" | tail -120 > gpurun_out/r2b_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 2>gpurun_out/r2b_bench.err | tail -1 > gpurun_out/r2b_bench_cfg5.json
tail -30 gpurun_out/r2b_tests.log
cut -c1-1500 gpurun_out/r2b_bench_cfg5.json; tail -5 gpurun_out/r2b_bench.err
