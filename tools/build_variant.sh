#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO=..]...  ->  tools/variants/libctd_NAME.so  (use with CTD_HIP_LIB=...)
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/variants/obj_$name
pids=()
for f in connecting_the_dots_amd/csrc/*.hip; do
  o=tools/variants/obj_$name/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude "$@" -c $f -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/variants/obj_$name/*.o -o tools/variants/libctd_$name.so && rm -rf tools/variants/obj_$name
