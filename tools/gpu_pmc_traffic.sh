#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes -> profiles/pmc_traffic.json (HBM-side bytes per launch of each entry point)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="--steps 10 --warmup 200 --no-cpu-baseline --keyframes 16"
cd /tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/r02_h_pmc_fetch_write.txt 2>&1
python tools/pmc_traffic.py gpurun_out/r02_h_pmc_traffic.json; cat gpurun_out/r02_h_pmc_traffic.json | cut -c1-900
tail -1 gpurun_out/pmc_write.log | cut -c1-120
find gpurun_out -name "*.db" -size +30M -delete
