#!/bin/bash
# One pass over variant libraries (timing-only ablations): tools/ab_once.sh lib...   -- prints tools/kb1.py's line per lib
for lib in "$@"; do
  echo -n "$(basename $lib): "
  UITK_LIB_PATH=$PWD/$lib timeout 200 python tools/kb1.py 2>&1 | tail -1
done
