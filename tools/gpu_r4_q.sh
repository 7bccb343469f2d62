#!/bin/bash
mkdir -p gpurun_out
export NOF_DIST_BACKEND=nccl HSA_ENABLE_IPC_MODE_LEGACY=0 NOF_DP_FORCE=1 NOF_DP_PAYLOAD=fp32 NOF_DP_MODE=allreduce NOF_BENCH_TRACE_STEPS=1 NOF_DP_OVERLAP=0
rm -rf /tmp/traces; mkdir -p /tmp/traces
for i in $(seq 1 16); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 1 --steps 6 --warmup 2 --keyframes 3 --no-cpu-baseline --settle 0 --round-steps 0 2>&1 >/dev/null | grep "^\[step" > /tmp/traces/trace_$i.txt
  echo "run $i: $(tail -1 /tmp/traces/trace_$i.txt | cut -c1-90)"
done
tar czf gpurun_out/r04_q_traces.tgz -C /tmp traces; ls -la gpurun_out/r04_q_traces.tgz
