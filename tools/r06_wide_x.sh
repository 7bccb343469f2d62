#!/bin/bash
# round 6: cfg5 bench only (no parity tests) for the regular library and A/B builds -- used for the wide backward's ablation, whose timing-only
# knobs (NOF_WIDE_X: results wrong by design) lived in nof_mlp_wide.hip at commit cabe347 and went with the pipelined rewrite that followed
#   gpurun -- 'bash tools/r06_wide_x.sh <tag> ab_x.so ...'
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r06_x}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${T}_build.txt 2>&1 || { tail -20 gpurun_out/${T}_build.txt; exit 1; }
CFG5="--mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 --no-cpu-baseline --no-extra-configs --settle 0 --round-steps 0 --steps 20 --warmup 10 --keyframes 8"
run() {
  echo "== $1"
  if [ "$1" != regular ]; then export NOF_LIB=$PWD/bundlesdf_amd/$1; else unset NOF_LIB; fi
  timeout 600 python bench.py $CFG5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_warmup']
print('ms/step', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'dense', round(d.get('ms_per_step_dense_backward') or 0,4), 'loss', d['loss'], '|', ' '.join(f'{n}={v:.4f}' for n,v in list(k.items())[:6]))"
}
{ run regular; for so in "$@"; do run $so; done; } 2>&1 | tee gpurun_out/${T}_wide_x.txt
