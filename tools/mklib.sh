#!/bin/bash
# Build a variant library for same-box A/B: tools/mklib.sh name [extra hipcc flags...]  ->  tools/ab/libuitk_<name>.so
name=$1; shift
cd "$(dirname "$0")/../uit_mobile_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-result -Wno-inline-asm "$@" \
  -o ../../tools/ab/libuitk_$name.so api.cpp pack.cpp frontend.hip encoder.hip epilogue.hip calibrate.hip 2>&1 | grep -E " error|spill" ; ls -la ../../tools/ab/libuitk_$name.so | awk '{print $5, $9}'
