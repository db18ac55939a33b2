#!/bin/bash
# Same-box A/B of library builds: AB_PRECS="f16 f16l6" tools/ab_l6.sh libA.so libB.so ...
for r in 1 2; do
  for lib in "$@"; do
    for prec in ${AB_PRECS:-f16 f16l6}; do
      echo -n "round $r $(basename $lib) $prec: "
      KB_PREC=$prec UITK_LIB_PATH=$PWD/$lib timeout 200 python tools/kb.py 2>&1 | tail -1
    done
  done
done
