#!/bin/bash
# round 2: whole GPU suite + smoke + default bench (with the CPU leg)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^z_in_out\|^ERROR sample" | tail -25 > gpurun_out/r2d_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2d_smoke.log 2>&1; tail -1 gpurun_out/r2d_smoke.log
timeout 400 python bench.py 2> gpurun_out/r2d_bench.err | tail -1 > gpurun_out/r2d_bench.json
tail -8 gpurun_out/r2d_tests.log; cut -c1-2500 gpurun_out/r2d_bench.json; tail -3 gpurun_out/r2d_bench.err
