#!/bin/bash
# round 2, after the scatter rewrite: PMC passes (atomics seen by the L2, HBM-side bytes, MFMA / VALU activity), the cfg5 bench line
# and the bench variants.  Counters in their own passes (no --stats, no other trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --keyframes 16"
cd /tmp
pass() { name=$1; shift; rm -rf $R/gpurun_out/pmc_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$name -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_$name.log 2>&1 || echo "pass $name failed: $(tail -2 $R/gpurun_out/pmc_$name.log)"; }
pass atomic TCC_ATOMIC_sum TCC_REQ_sum
pass ea TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES
cd $R
python tools/pmc_summary.py gpurun_out/pmc_atomic gpurun_out/pmc_ea gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_valu > gpurun_out/r02_e_pmc_counters.txt 2>&1
python tools/pmc_traffic.py gpurun_out/r02_e_pmc_traffic.json > /dev/null 2>&1
grep -i "agg\|k_mlp_bwd\|k_mlp_fwd\|k_hash_fwd\|k_hash_dx" gpurun_out/r02_e_pmc_counters.txt | head -60
find gpurun_out -name "*.db" -size +30M -delete
rm -f gpurun_out/r02_e_bench_variants.jsonl
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 2>/dev/null | tail -1 > gpurun_out/r02_e_bench_cfg5.json
cut -c1-200 gpurun_out/r02_e_bench_cfg5.json
for p in bf16 fp16 bf16x3; do timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --precision $p 2>/dev/null | tail -1 >> gpurun_out/r02_e_bench_variants.jsonl; done
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --mlp reference 2>/dev/null | tail -1 >> gpurun_out/r02_e_bench_variants.jsonl
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --rays 8192 2>/dev/null | tail -1 >> gpurun_out/r02_e_bench_variants.jsonl
cut -c1-200 gpurun_out/r02_e_bench_variants.jsonl
