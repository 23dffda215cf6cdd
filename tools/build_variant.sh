#!/bin/bash
# tools/build_variant.sh NAME [FILE.hip] [-DMACRO=..]...  ->  tools/variants/libctd_NAME.so
# An experimental build of the library for A/B timing with tools/time_variant.py (never shipped: tools/variants/ is
# git-ignored).  FILE.hip replaces the csrc file of the same base name (default: csrc/ncc_fast.hip as it is in the
# tree); every other object comes from the regular build (build/obj, run `python -m connecting_the_dots_amd.build` first).
name=$1; shift
cd "$(dirname "$0")/.."
src=connecting_the_dots_amd/csrc/ncc_fast.hip
if [ -n "$1" ] && [ "${1:0:1}" != "-" ]; then src=$1; shift; fi
base=$(basename $src .hip); base=${base%%@*}
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -Iconnecting_the_dots_amd/csrc "$@" -c $src -o tools/variants/$name.o || exit 1
objs=$(ls build/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/variants/$name.o -o tools/variants/libctd_$name.so && rm -f tools/variants/$name.o
