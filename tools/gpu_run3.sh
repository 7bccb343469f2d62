#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_runner.py tests/test_gpu_dp.py tests/test_gpu_texture.py -q -x -s 2>&1 | grep -v "^chamfer\|^sc_factor\|^translation\|^rays" | tail -40 > gpurun_out/r03_d_tests.txt; tail -30 gpurun_out/r03_d_tests.txt
