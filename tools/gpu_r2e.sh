#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for w in 1 2 3; do
  NOF_AGG_WGS=$w timeout 300 python tools/hash_bwd_probe.py 2>&1 | grep "whole call" | sed "s/default/wgs=$w/"
  NOF_AGG_WGS=$w timeout 300 python bench.py --steps 100 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step', d['ms_per_step'], 'captured', d['captured_step_ms_per_step'], 'bwd avg', d['roofline']['avg_ms'])"
done
} | tee gpurun_out/r02_hash_probe_v4c.txt
