#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh libA.so libB.so [...]   (run on the GPU box via gpurun)
# Alternates the builds over 3 rounds in separate processes and prints per-kernel medians.
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "round $r $(basename $lib): "
    UITK_LIB_PATH=$PWD/$lib timeout 200 python tools/kb1.py 2>&1 | tail -1
  done
done
