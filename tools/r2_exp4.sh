#!/bin/bash
# structured rounding / split patch embed: parity (seed sweep) and speed per configuration, same box
mkdir -p gpurun_out/r2_exp4
for cfg in "UITK_SPLIT_PE=1" "UITK_SPLIT_PE=0" "UITK_SPLIT_PE=0 UITK_PACK_RTN=1" "UITK_SPLIT_PE=1 UITK_PACK_RTN=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 900 python tools/seed_sweep.py 6 2>&1 | grep -v amdgpu > gpurun_out/r2_exp4/sweep_$tag.log
  echo "== $cfg"; grep -E "trained|scale 3|uit_xs    scale 1" gpurun_out/r2_exp4/sweep_$tag.log
  env $cfg KB_PREC=f16 timeout 200 python tools/kb_fp8.py 2>&1 | tail -1
done
