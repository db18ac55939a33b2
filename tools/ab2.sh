#!/bin/bash
# Same-box A/B of (library, environment) pairs: tools/ab2.sh "lib.so VAR=val ..." "lib2.so ..."   (run on the GPU box)
specs=("$@")
for r in 1 2 3; do
  for spec in "${specs[@]}"; do
    read -r -a parts <<< "$spec"
    lib=${parts[0]}
    echo -n "round $r [$spec]: "
    env UITK_LIB_PATH=$PWD/$lib "${parts[@]:1}" timeout 200 python tools/kb1.py 2>&1 | tail -1
  done
done
