#!/bin/bash
# int16-input bench line of several library builds on one box: tools/ab_i16.sh libA.so libB.so ...
for r in 1 2; do for lib in "$@"; do
  echo -n "round $r $(basename $lib): "
  UITK_LIB_PATH=$PWD/$lib timeout 200 python bench.py --pcm16 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s  frontend', d['frontend']['launch_ms'], 'ms')"
done; done
