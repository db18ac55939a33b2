#!/bin/bash
# Same-box A/B of two source trees (each with its own built lib): tools/ab_tree.sh dirA dirB
for r in 1 2 3; do
  for d in "$@"; do
    echo -n "round $r $d: "
    (cd $d && timeout 200 python tools/kb1.py 2>&1 | tail -1)
  done
done
