#!/bin/bash
# Lab build that differs from the product in encoder.hip only: tools/mkenc.sh name [extra hipcc flags...]  ->  tools/ab/libuitk_<name>.so
# (re-uses the product's other objects under uit_mobile_amd/csrc/build/: run `python -m uit_mobile_amd.build` first)
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/uit_mobile_amd/csrc" || exit 1
mkdir -p "$root/tools/ab"
obj="$root/tools/ab/encoder_$name.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -Wno-inline-asm -DUITK_LAB "$@" -c -o "$obj" encoder.hip 2>&1 | grep -E " error|spill"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libuitk_$name.so" build/api.cpp.o build/pack.cpp.o build/frontend.hip.o build/epilogue.hip.o build/calibrate.hip.o "$obj" && rm -f "$obj"
ls -la "$root/tools/ab/libuitk_$name.so" | awk '{print $5, $9}'
