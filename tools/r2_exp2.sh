#!/bin/bash
# round 2, run 2: GPU tests after the parity work + seed sweep (trained-like) + bench sanity
mkdir -p gpurun_out/r2_exp2
timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r2_exp2/pytest.log
timeout 900 python tools/seed_sweep.py 6 > gpurun_out/r2_exp2/seed_sweep.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_exp2/bench.log 2>&1
cat gpurun_out/r2_exp2/pytest.log; tail -12 gpurun_out/r2_exp2/seed_sweep.log; tail -2 gpurun_out/r2_exp2/bench.log
