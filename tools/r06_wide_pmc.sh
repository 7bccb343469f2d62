#!/bin/bash
# round 6: where the wide backward's wave time goes -- SQ counter passes at cfg5 (separate --pmc runs, kernel trace only)
#   gpurun -- 'bash tools/r06_wide_pmc.sh <tag>'
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; T=${1:-r06_c}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0 --steps 4 --warmup 30 --keyframes 8"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); d=$R/gpurun_out/pmc_w$i; rm -rf $d
  timeout 500 rocprofv3 --pmc $C --kernel-trace -d $d -o b -- python $R/bench.py $CFG5 > $d.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_w1 $R/gpurun_out/pmc_w2 $R/gpurun_out/pmc_w3 > $R/gpurun_out/${T}_pmc_wide.txt 2>&1
grep "k_wide\|==" $R/gpurun_out/${T}_pmc_wide.txt | cut -c1-130
find $R/gpurun_out -name "*.db" -delete; rm -rf $R/gpurun_out/pmc_w1 $R/gpurun_out/pmc_w2 $R/gpurun_out/pmc_w3
