#!/bin/bash
# Build-container helper of the A/B experiments: a small-instance-only library with extra compiler flags.
#   bash tools/exp/build_variant.sh NAME [extra hipcc flags ...]   ->   .ab/r06/NAME/libcookingzoo_hip.so   (CZ_LIB=... / tools/ab_quick.sh LIBS=...)
set -e
cd "$(dirname "$0")/../../cooking_zoo_amd/csrc"
name=${1:?variant name}; shift
d=../../.ab/r06/$name
mkdir -p "$d"
for u in cz_api cz_inst_small; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=14 \
      -DCZ_SMALL_ONLY "$@" -c -o "$d/$u.o" $u.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$d/libcookingzoo_hip.so" "$d/cz_api.o" "$d/cz_inst_small.o" -ldl
rm -f "$d/cz_api.o" "$d/cz_inst_small.o"
ls -la "$d/libcookingzoo_hip.so"
