#!/bin/bash
# round-end evidence in one gpurun call; only text summaries are left under gpurun_out/ (the merge back is capped at 64 MiB)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_* gpurun_out/prof* 
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/tests_all.log 2>&1; tail -3 gpurun_out/tests_all.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 420 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-300
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; python tools/prof_summary.py gpurun_out/prof/bench_results.db > gpurun_out/kernel_stats.txt 2>&1; head -12 gpurun_out/kernel_stats.txt | cut -c1-120
KF=16 bash tools/gpu_pmc.sh > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_traffic.json
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1
timeout 200 python tools/extract_bench.py 2>/dev/null | grep "^{" > gpurun_out/extract.jsonl; cat gpurun_out/extract.jsonl | cut -c1-200
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_tcc gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/prof gpurun_out/prof2 gpurun_out/prof3
ls -la gpurun_out | head -30
