#!/bin/bash
# round 2, experiment 1: two-stream overlap of front-end and encoder (same box, alternating)
mkdir -p gpurun_out/r2_exp1
L=gpurun_out/r2_exp1/overlap.log
: > $L
for e in "UITK_ENC_WAVES=8 UITK_FE_WG_PER_CU=2" "UITK_ENC_WAVES=4 UITK_FE_WG_PER_CU=1" "UITK_ENC_WAVES=4 UITK_FE_WG_PER_CU=2" "UITK_ENC_WAVES=8 UITK_FE_WG_PER_CU=1" "UITK_ENC_WAVES=4 UITK_FE_WG_PER_CU=1 OB_REPLICAS=3"; do
  env $e timeout 300 python tools/overlap_bench.py 200 >> $L 2>&1
done
timeout 900 python tools/seed_sweep.py 4 > gpurun_out/r2_exp1/seed_sweep_r1_state.log 2>&1
cat $L
