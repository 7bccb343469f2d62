#!/bin/bash
# A/B of two builds (ab_old.so / ab_new.so) with a kernel trace each: per-kernel averages of the MLP kernels + parity of the new one
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in old new; do
  cp ab_$v.so bundlesdf_amd/libnof_hip.so
  rm -rf /tmp/prof_$v; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline > /tmp/prof_$v.json 2>/dev/null)
  echo "== $v $(python -c "import json; d=json.loads(open('/tmp/prof_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  python tools/prof_summary.py $(find /tmp/prof_$v -name "*_results.db" | head -1) | grep "k_mlp\|k_hash" | cut -c1-60,73-110
done

