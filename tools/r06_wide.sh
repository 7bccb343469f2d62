#!/bin/bash
# round 6: the on-chip wide backward -- parity tests of the wide path, then a short cfg5 bench + kernel trace + timeline
#   gpurun -- 'bash tools/r06_wide.sh <tag> [pytest -k expression]'
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; T=${1:-r06_a}; K=${2:-wide}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_erratum.py tests/test_gpu_ops.py tests/test_gpu_tiles.py tests/test_gpu_step.py -q -k "$K" -s 2>&1 | tail -80 > gpurun_out/${T}_wide_tests.txt
tail -30 gpurun_out/${T}_wide_tests.txt | cut -c1-400
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0"
timeout 600 python bench.py $CFG5 --steps 20 --warmup 10 --keyframes 16 2> gpurun_out/${T}_bench_cfg5.log | tail -1 > gpurun_out/${T}_bench_cfg5.json
python - <<PY
import json
d = json.load(open('gpurun_out/${T}_bench_cfg5.json'))
print({k: d.get(k) for k in ('ms_per_step', 'value', 'loss', 'flags', 'step_ms_spread', 'kernel_ms_warmup')})
PY
cd /tmp
rm -rf $R/gpurun_out/prof_5; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_5 -o b -- python $R/bench.py $CFG5 --steps 20 --warmup 30 --keyframes 8 > $R/gpurun_out/${T}_trace_bench_cfg5.json 2>$R/gpurun_out/${T}_trace_cfg5.log
db=$(find $R/gpurun_out/prof_5 -name "*.db" | head -1)
python $R/tools/prof_summary.py $db > $R/gpurun_out/${T}_cfg5_kernel_stats.txt 2>&1
python $R/tools/step_timeline.py $db 40 > $R/gpurun_out/${T}_cfg5_timeline.txt 2>&1
head -22 $R/gpurun_out/${T}_cfg5_kernel_stats.txt | cut -c1-60,73-150; cut -c1-110 $R/gpurun_out/${T}_cfg5_timeline.txt
find $R/gpurun_out -name "*.db" -delete; rm -rf $R/gpurun_out/prof_5
