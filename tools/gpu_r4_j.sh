#!/bin/bash
# kernel trace + two-step timeline of the settled step (deferred table Adam on / off)
tag=${1:-r04_j}
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
for d in 1 0; do
  rm -rf $R/gpurun_out/prof_t
  NOF_DEFER_ADAM=$d timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t -o b -- python $R/bench.py --steps 60 --warmup 200 --settle 0 --round-steps 0 --preroll 0 --no-cpu-baseline --no-extra-configs --keyframes 16 > $R/gpurun_out/${tag}_trace_bench_defer$d.json 2>$R/gpurun_out/${tag}_trace_defer$d.log
  db=$(find $R/gpurun_out/prof_t -name "*.db" | head -1)
  python $R/tools/step_timeline.py $db 240 > $R/gpurun_out/${tag}_timeline_defer$d.txt 2>&1
  python $R/tools/step_timeline.py $db 241 >> $R/gpurun_out/${tag}_timeline_defer$d.txt 2>&1
  rm -rf $R/gpurun_out/prof_t
done
cat $R/gpurun_out/${tag}_timeline_defer1.txt | cut -c1-110
