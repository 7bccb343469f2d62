#!/bin/bash
# the figures the parity tests print (run with -s): HIP vs the reference-driven fixture, the full-size steps, the wide step, the renderer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_reference_fixture.py tests/test_gpu_fullsize.py::test_fullsize_step_matches_oracle "tests/test_gpu_step.py::test_wide_network_step_matches_oracle" tests/test_gpu_step.py::test_default_precision_meets_1e3_on_outputs tests/test_gpu_render.py -q -s --timeout=400 -p no:cacheprovider 2>&1 | grep -v "^sc_factor\|^translation\|^rays \|Octree\|amdgpu.ids" | grep "fullsize\|fp32:\|fp16x3:\|bf16x3:\|wide 4x128\|passed\|failed\|colour\|default\|render " > gpurun_out/${1:-r04}_parity_numbers.txt; cat gpurun_out/${1:-r04}_parity_numbers.txt
