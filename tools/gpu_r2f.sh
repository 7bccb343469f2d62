#!/bin/bash
# round 2, after the scatter rewrite: whole GPU suite, smoke, bench (+ kernel trace), atomic probe record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r02_d_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2>gpurun_out/r02_d_bench.log | tail -1 > gpurun_out/r02_d_bench.json; cut -c1-300 gpurun_out/r02_d_bench.json
python tools/atomic_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_d_atomic_probe.txt
rm -rf gpurun_out/prof_d; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d -o bench -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r02_d_prof_bench.json 2>gpurun_out/r02_d_prof.log
python tools/prof_summary.py $(find gpurun_out/prof_d -name "*_results.db" | head -1) > gpurun_out/r02_d_kernel_stats.txt 2>&1; head -12 gpurun_out/r02_d_kernel_stats.txt
find gpurun_out/prof_d -name "*.db" -size +60M -delete
