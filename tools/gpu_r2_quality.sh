#!/bin/bash
# round 2: converged-field quality per operand precision (same seed), the cfg5 bench line, the plain 16-bit bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r02_quality_cfg2.jsonl gpurun_out/r02_bench_variants.jsonl
for p in fp32 fp16x3 fp16 bf16; do
  timeout 300 python tools/quality_cfg2.py --precision $p 2>/dev/null | grep "^{" >> gpurun_out/r02_quality_cfg2.jsonl
done
cat gpurun_out/r02_quality_cfg2.jsonl | cut -c1-420
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mlp cfg5 --rays 16384 --log2_T 22 --finest 512 --width 1280 --height 720 --precision fp16 2>/dev/null | tail -1 > gpurun_out/r02_bench_cfg5.json
cut -c1-600 gpurun_out/r02_bench_cfg5.json
for p in bf16 fp16 bf16x3; do timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --precision $p 2>/dev/null | tail -1 >> gpurun_out/r02_bench_variants.jsonl; done
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --mlp reference 2>/dev/null | tail -1 >> gpurun_out/r02_bench_variants.jsonl
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --rays 8192 2>/dev/null | tail -1 >> gpurun_out/r02_bench_variants.jsonl
cut -c1-200 gpurun_out/r02_bench_variants.jsonl
