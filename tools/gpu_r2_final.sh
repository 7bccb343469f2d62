#!/bin/bash
# round 2, final evidence: bench with the CPU leg, kernel trace of the same command, MFMA-busy PMC pass
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py 2>gpurun_out/r02_h_bench.log | tail -1 > gpurun_out/r02_h_bench.json; cut -c1-240 gpurun_out/r02_h_bench.json
cd /tmp
rm -rf $R/gpurun_out/prof_h; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_h -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r02_h_prof_bench.json 2>$R/gpurun_out/r02_h_prof.log
rm -rf $R/gpurun_out/pmc_mfma; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --keyframes 16 > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_h -name "*_results.db" | head -1) > gpurun_out/r02_h_kernel_stats.txt 2>&1; head -14 gpurun_out/r02_h_kernel_stats.txt | cut -c1-60,73-112
python tools/pmc_summary.py gpurun_out/pmc_mfma > gpurun_out/r02_h_pmc_mfma.txt 2>&1; grep "k_mlp" gpurun_out/r02_h_pmc_mfma.txt | grep "MFMA\|GRBM" | cut -c1-120
find gpurun_out -name "*.db" -size +30M -delete
