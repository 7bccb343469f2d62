#!/bin/bash
# The round's GPU evidence in one gpurun call:  gpurun -- 'bash tools/gpu_evidence.sh <tag> [legs]'
# legs (default "bench trace mfma traffic cfg5"):
#   bench    bench.py as the driver runs it (defaults, CPU leg included)                      -> <tag>_bench.json
#   trace    rocprofv3 --kernel-trace --stats of a settled run (200 warm-up steps) + timeline  -> <tag>_kernel_stats.txt, <tag>_timeline.txt
#   mfma     --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_WAVES / GRBM_GUI_ACTIVE  -> <tag>_pmc_mfma.txt
#   traffic  --pmc passes FETCH_SIZE, WRITE_SIZE (separate), TCC atomics                       -> <tag>_pmc_fetch_write.txt, <tag>_pmc_traffic.json
#   sq       --pmc pass: SQ wave-time breakdown                                                -> <tag>_pmc_sq.txt
#   vmem     --pmc passes: vector-memory / LDS / scalar instruction counts; L2 hits and misses (fused and two-launch forward) -> <tag>_pmc_vmem.txt
#   calib    --pmc FETCH_SIZE / WRITE_SIZE on a 146 MB float4 copy and on Adam over 36.5 M parameters  -> <tag>_pmc_calibration.txt
#   cfg5     bench + kernel trace + timeline at BASELINE cfg5                                  -> <tag>_bench_cfg5.json, <tag>_cfg5_*.txt
# PMC passes never share a rocprofv3 run with --stats / sys traces (gpurun refuses that); the databases are deleted once summarised (gpurun copies at most 64 MiB back).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }   # never measure a stale library
T=${1:-r04}; LEGS=${2:-"bench trace mfma traffic vmem calib cfg5"}
export KF=${KF:-64}                      # keyframes of the pool in the trace / counter legs (round 6: the headline's own 64; rounds 3-5 used 16)
SETTLED="--steps 60 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF"
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0"
has() { [[ " $LEGS " == *" $1 "* ]]; }
db() { find "$1" -name "*.db" | head -1; }
if has bench; then
  timeout 900 python bench.py 2>gpurun_out/${T}_bench.log | tail -1 > gpurun_out/${T}_bench.json; cut -c1-220 gpurun_out/${T}_bench.json
fi
(cd $R && python tools/kernel_metadata.py bundlesdf_amd/libnof_hip.so > gpurun_out/${T}_kernel_metadata.txt 2>&1)
cd /tmp
if has trace; then
  rm -rf $R/gpurun_out/prof_t; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t -o b -- python $R/bench.py $SETTLED > $R/gpurun_out/${T}_trace_bench.json 2>$R/gpurun_out/${T}_trace.log
  python $R/tools/prof_summary.py $(db $R/gpurun_out/prof_t) > $R/gpurun_out/${T}_kernel_stats.txt 2>&1; head -24 $R/gpurun_out/${T}_kernel_stats.txt | cut -c1-60,73-112
  python $R/tools/step_timeline.py $(db $R/gpurun_out/prof_t) 240 > $R/gpurun_out/${T}_timeline.txt 2>&1
fi
if has mfma; then
  rm -rf $R/gpurun_out/pmc_mfma; timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 10 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF > $R/gpurun_out/pmc_mfma.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_mfma > $R/gpurun_out/${T}_pmc_mfma.txt 2>&1; grep "k_mlp\|k_enc_mlp" $R/gpurun_out/${T}_pmc_mfma.txt | grep "MFMA\|GRBM" | cut -c1-120
fi
if has traffic; then
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1); rm -rf $d
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d $d -o b -- python $R/bench.py --steps 10 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF > $d.log 2>&1
  done
  rm -rf $R/gpurun_out/pmc_atomic; timeout 400 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum --kernel-trace -d $R/gpurun_out/pmc_atomic -o b -- python $R/bench.py --steps 10 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF > $R/gpurun_out/pmc_atomic.log 2>&1
  (cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_atomic > gpurun_out/${T}_pmc_fetch_write.txt 2>&1; python tools/pmc_traffic.py gpurun_out/${T}_pmc_traffic.json ${T}_pmc_fetch_write.txt | cut -c1-400)
fi
if has sq; then
  rm -rf $R/gpurun_out/pmc_sq; timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_sq -o b -- python $R/bench.py --steps 6 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF > $R/gpurun_out/pmc_sq.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_sq > $R/gpurun_out/${T}_pmc_sq.txt 2>&1
fi
if has vmem; then
  P="--steps 6 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes $KF"
  rm -rf $R/gpurun_out/pmc_v1 $R/gpurun_out/pmc_v2 $R/gpurun_out/pmc_v3
  timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_v1 -o b -- python $R/bench.py $P > $R/gpurun_out/pmc_v1.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/pmc_v2 -o b -- python $R/bench.py $P > $R/gpurun_out/pmc_v2.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/pmc_v3 -o b -- python $R/bench.py $P --unfused > $R/gpurun_out/pmc_v3.log 2>&1
  (echo "# pmc_v1: instruction counts per launch; pmc_v2: L2 (TCC) hits / misses, fused forward; pmc_v3: the same, two-launch forward (--unfused)"; python $R/tools/pmc_summary.py $R/gpurun_out/pmc_v1 $R/gpurun_out/pmc_v2 $R/gpurun_out/pmc_v3) > $R/gpurun_out/${T}_pmc_vmem.txt 2>&1
  grep "k_enc_mlp\|k_hash_fwd\|k_hash_dx\|k_mlp_fwd" $R/gpurun_out/${T}_pmc_vmem.txt | cut -c1-130 | head -40
fi
if has calib; then
  rm -rf $R/gpurun_out/pmc_c1 $R/gpurun_out/pmc_c2
  (cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_c1 -o b -- python tools/copy_probe.py > gpurun_out/pmc_c1.log 2>&1; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_c2 -o b -- python tools/copy_probe.py > gpurun_out/pmc_c2.log 2>&1)
  (echo "# 146 000 000 bytes copied per launch (fetch 146 MB, write 146 MB); Adam over 36.5 M parameters: 584 MB read + 584 MB written per launch (16 B + 16 B per parameter); counter values are KiB"; python $R/tools/pmc_summary.py $R/gpurun_out/pmc_c1 $R/gpurun_out/pmc_c2) > $R/gpurun_out/${T}_pmc_calibration.txt 2>&1
  cat $R/gpurun_out/${T}_pmc_calibration.txt | cut -c1-130 | head -20
fi
if has cfg5; then
  (cd $R && timeout 900 python bench.py $CFG5 --steps 40 --warmup 100 --keyframes 16 2>gpurun_out/${T}_bench_cfg5.log | tail -1 > gpurun_out/${T}_bench_cfg5.json; cut -c1-220 gpurun_out/${T}_bench_cfg5.json)
  rm -rf $R/gpurun_out/prof_5; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_5 -o b -- python $R/bench.py $CFG5 --steps 20 --warmup 100 --keyframes 8 > $R/gpurun_out/${T}_trace_bench_cfg5.json 2>$R/gpurun_out/${T}_trace_cfg5.log
  python $R/tools/prof_summary.py $(db $R/gpurun_out/prof_5) > $R/gpurun_out/${T}_cfg5_kernel_stats.txt 2>&1
  python $R/tools/step_timeline.py $(db $R/gpurun_out/prof_5) 105 > $R/gpurun_out/${T}_cfg5_timeline.txt 2>&1; grep "k_wide\|k_hash\|k_adam" $R/gpurun_out/${T}_cfg5_timeline.txt | cut -c1-100
fi
find $R/gpurun_out -name "*.db" -delete; rm -rf $R/gpurun_out/prof_t $R/gpurun_out/prof_5 $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_atomic $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_v1 $R/gpurun_out/pmc_v2 $R/gpurun_out/pmc_v3 $R/gpurun_out/pmc_c1 $R/gpurun_out/pmc_c2; du -sh $R/gpurun_out
