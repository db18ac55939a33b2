#!/bin/bash
# Same-box A/B of pack/run-time knobs with tools/kb1.py: tools/ab_env_kb1.sh "VAR=val ..." "VAR=val ..."
for r in 1 2 3; do
  for e in "$@"; do
    echo -n "round $r [$e]: "
    env $e timeout 200 python tools/kb1.py 2>&1 | tail -1
  done
done
