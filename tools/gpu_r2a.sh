#!/bin/bash
# round 2, first GPU pass: the whole GPU suite + the bench in the new default precision and the plain 16-bit modes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | tail -150 > gpurun_out/r2a_tests.log
for p in fp16x3 bf16 fp16 bf16x3; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --precision $p 2>>gpurun_out/r2a_bench.err | tail -1 >> gpurun_out/r2a_bench.jsonl
done
tail -5 gpurun_out/r2a_tests.log
cat gpurun_out/r2a_bench.jsonl | cut -c1-400
