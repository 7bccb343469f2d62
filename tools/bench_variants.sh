#!/bin/bash
# bench.py on the other BASELINE-relevant single-GPU shapes (short runs, no CPU leg)
cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'args': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'it_s': d['train_iters_per_sec'], 'dtype': d['dtype'], 'workload': d['config']['workload'][:60]}))"; }
run --precision fp16
run --mlp reference
run --mlp reference --precision fp16
run --rays 8192
run --precision fp32 --keyframes 16
