#!/bin/bash
# lab: precision f16l6 with and without the split patch embed (error on the worst sweep seeds + kernel time)
for s in 1 0; do
  echo "== UITK_SPLIT_PE=$s"
  UITK_SPLIT_PE=$s python tools/l6_check.py uit_xs 205 207 2>&1 | grep f16l6
  UITK_SPLIT_PE=$s KB_PREC=f16l6 python tools/kb_fp8.py 2>&1 | tail -1
done
