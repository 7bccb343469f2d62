#!/bin/bash
# round-2 evidence in one gpurun call: kernel trace of the default bench, PMC passes (separate from the trace: FETCH_SIZE,
# WRITE_SIZE, MFMA-busy, TCC hit/miss, SQ wave time), traffic json; only text summaries are left under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_* gpurun_out/prof*
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; python tools/prof_summary.py gpurun_out/prof/bench_results.db > gpurun_out/r02_kernel_stats.txt 2>&1; head -24 gpurun_out/r02_kernel_stats.txt | cut -c1-150
KF=16 bash tools/gpu_pmc.sh > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r02_pmc_counters.txt
python tools/pmc_traffic.py gpurun_out/r02_pmc_traffic.json
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1
cp gpurun_out/pmc_sq_summary.txt gpurun_out/r02_pmc_sq_wave_time.txt
timeout 200 python tools/extract_bench.py 2>/dev/null | grep "^{" > gpurun_out/r02_extract_512.jsonl; cut -c1-200 gpurun_out/r02_extract_512.jsonl
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_tcc gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/prof
grep -E "mlp|hash" gpurun_out/r02_pmc_counters.txt | head -30 | cut -c1-160
