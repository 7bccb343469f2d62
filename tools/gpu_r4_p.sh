#!/bin/bash
# the flaky one-rank RCCL test, repeated: which segment of the flat buffer do the runs disagree on?
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  python -m pytest tests/test_gpu_dp.py::test_rccl_calls_of_the_bucketed_step_one_rank -x -q -m gpu -s 2>&1 | grep -E "passed|failed|assert .*<=|table|E  " | cut -c1-700
done > gpurun_out/r04_p_rccl_repeat.txt
cat gpurun_out/r04_p_rccl_repeat.txt
