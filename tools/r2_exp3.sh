#!/bin/bash
mkdir -p gpurun_out/r2_exp3
bash tools/ab_prec.sh tools/ab/libuitk_cur.so tools/ab/libuitk_pesplit.so 2>&1 | grep -v amdgpu
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|trained-like|fp8 |Error" 
timeout 900 python tools/seed_sweep.py 6 2>&1 | grep -v amdgpu > gpurun_out/r2_exp3/seed_sweep_split_pe.log; grep trained gpurun_out/r2_exp3/seed_sweep_split_pe.log; grep "scale 3" gpurun_out/r2_exp3/seed_sweep_split_pe.log
