#!/bin/bash
# static instruction / branch counts of cz::k_step<1,1,2,3,0> under extra compiler flags (build container; ~25 s per call)
#   bash tools/exp/static_counts.sh [extra hipcc flags ...]
cd "$(dirname "$0")/../../cooking_zoo_amd/csrc"
out=$(mktemp /tmp/cz_small_XXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=14 \
    -DCZ_SMALL_ONLY "$@" -S --cuda-device-only -o "$out" cz_inst_small.hip 2>/dev/null || { echo "compile failed"; exit 1; }
awk '/^_ZN2cz6k_stepILi1ELi1ELi2ELi3ELi0EEE.*:/{on=1} on&&/s_endpgm/{print; on=0} on{print}' "$out" > "$out.k"
echo "flags: $*"
echo "  instructions $(grep -cE '^\s+(s_|v_|ds_|buffer_|global_|flat_)' "$out.k")  branches $(grep -cE '^\s+s_cbranch|^\s+s_branch' "$out.k")  cbranch_exec $(grep -cE 's_cbranch_exec' "$out.k")  cbranch_scc/vcc $(grep -cE 's_cbranch_(scc|vcc)' "$out.k")  v_cndmask $(grep -c v_cndmask "$out.k")  s_cselect $(grep -c s_cselect "$out.k")  readlane $(grep -cE 'v_readlane|v_readfirstlane' "$out.k")"
grep -A12 "k_stepILi1ELi1ELi2ELi3ELi0EEE" "$out" | grep -E "sgpr_count|vgpr_count|spill" | head -4
rm -f "$out" "$out.k"
