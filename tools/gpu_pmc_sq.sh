#!/bin/bash
# SQ wave-time breakdown (quad-cycles): where the waves of each kernel spend their time
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --keyframes ${KF:-16}"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_sq -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace -d $R/gpurun_out/pmc_sq2 -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_sq2 > gpurun_out/pmc_sq_summary.txt 2>&1
grep -E "agg|mlp_bwd|hash_fwd|hash_dx|bwd_lds|^==" gpurun_out/pmc_sq_summary.txt | cut -c1-130
