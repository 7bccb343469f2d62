#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
python -m pytest tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -2
done > gpurun_out/r04_r_dp_repeat.txt
cat gpurun_out/r04_r_dp_repeat.txt
