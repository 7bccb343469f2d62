#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python tools/texture_edge_probe.py > gpurun_out/r06_q_texture_probe.txt 2>&1; head -40 gpurun_out/r06_q_texture_probe.txt | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_rays.py tests/test_gpu_dp.py "tests/test_gpu_ops.py::test_mlp_wide_forward_backward" "tests/test_gpu_fullsize.py::test_fullsize_step_matches_oracle" -q -x -s --timeout=900 -p no:cacheprovider -k "rays or dp or wide or cfg5" --durations=8 2>&1 | grep -v "^sc_factor\|^translation\|^rays \|Octree\|amdgpu.ids\|^wide\|^\.wide" | tail -40 > gpurun_out/r06_q_tests.txt; tail -32 gpurun_out/r06_q_tests.txt | cut -c1-260
