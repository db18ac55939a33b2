#!/bin/bash
# Same-box A/B of runtime knobs: tools/ab_env.sh "VAR=val ..." "VAR=val ..."   (each arg = one environment)
for r in 1 2 3; do
  for e in "$@"; do
    echo -n "round $r [$e]: "
    env $e timeout 200 python tools/kb1.py 2>&1 | tail -1
  done
done
