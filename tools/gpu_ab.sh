#!/bin/bash
# A/B of two builds of libnof_hip.so on the same box: ab_old.so / ab_new.so in the repo root
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in old new; do
  cp ab_$v.so bundlesdf_amd/libnof_hip.so
  timeout 600 python bench.py --no-cpu-baseline --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']; print('$v', round(d['ms_per_step'],4), round(d['captured_step_ms_per_step'],4), 'scatter', round(d['roofline']['avg_ms'],4))"
done
done
