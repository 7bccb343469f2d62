#!/bin/bash
# PMC passes for the bench (separate from --kernel-trace/--stats runs; FETCH_SIZE and WRITE_SIZE need separate passes:
# TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots")
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --keyframes ${KF:-16}"
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_mfma.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/pmc_tcc -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_tcc.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_tcc > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt | head -60
