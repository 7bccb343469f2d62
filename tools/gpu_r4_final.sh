#!/bin/bash
# round 4, final state: the whole GPU suite (with the figures the parity tests print), the driver's bench invocation, kernel trace + timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_final}
timeout 1700 python -m pytest tests -m gpu -q -s --durations=12 --timeout=600 -p no:cacheprovider > gpurun_out/${T}_gpu_tests_full.txt 2>&1
grep -v "^sc_factor\|^translation\|^rays \|Octree\|amdgpu.ids" gpurun_out/${T}_gpu_tests_full.txt | grep "fullsize\|fp32:\|fp16x3:\|bf16x3:\|wide 4x128\|colour\|default\|render \|query_network\|one-rank RCCL\|loss after" | cut -c1-400 > gpurun_out/${T}_parity_numbers.txt
tail -30 gpurun_out/${T}_gpu_tests_full.txt > gpurun_out/${T}_gpu_tests.txt; rm -f gpurun_out/${T}_gpu_tests_full.txt
tail -6 gpurun_out/${T}_gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench_driver.log | tail -1 > gpurun_out/${T}_bench_driver_invocation.json; cut -c1-330 gpurun_out/${T}_bench_driver_invocation.json
R=$(pwd) bash tools/gpu_evidence.sh $T trace
