#!/bin/bash
# Same-box A/B of runtime knobs with a chosen script: tools/ab_env2.sh script.py "VAR=val ..." "VAR=val ..."
S=$1; shift
for r in 1 2; do
  for e in "$@"; do
    echo -n "round $r [$e]: "
    env $e timeout 200 python $S 2>&1 | tail -1
  done
done
