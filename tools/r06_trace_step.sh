#!/bin/bash
# kernel trace of a short bench run + the timeline of chosen steps:  bash tools/r06_trace_step.sh <tag> "<bench args>" <step> [<step> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; T=$1; ARGS=$2; shift 2
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
cd /tmp; rm -rf $R/gpurun_out/prof_x
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o b -- python $R/bench.py $ARGS > $R/gpurun_out/${T}_bench.json 2>$R/gpurun_out/${T}_trace.log
DB=$(find $R/gpurun_out/prof_x -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB > $R/gpurun_out/${T}_kernel_stats.txt 2>&1
for s in "$@"; do python $R/tools/step_timeline.py $DB $s; done | tee $R/gpurun_out/${T}_timeline.txt
rm -rf $R/gpurun_out/prof_x
