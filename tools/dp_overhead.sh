#!/bin/bash
# cost of the data-parallel code path itself: bench.py through torch.distributed.run at world size 1 (RCCL), bucketed
# asynchronous all-reduce vs one blocking all-reduce
cd $GRAFT_REPO_ROOT
for o in 1 0; do
  NOF_DP_OVERLAP=$o HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 2951$o bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/dp_$o.json
  python - <<PY
import json
d = json.load(open('gpurun_out/dp_$o.json'))
print('overlap=$o', d['ms_per_step'], {k: v for k, v in d['kernel_ms_warmup'].items() if 'hash' in k or 'adam' in k})
PY
done
