for f in "-DCZ_CHAIN_WPE=0" "-DCZ_CHAIN_WPE=8" "-DCZ_CHAIN_WPE=0" "-DCZ_CHAIN_WPE=8"; do
  (cd cooking_zoo_amd/csrc && make clean >/dev/null && make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=14 $f" >/dev/null 2>&1)
  echo "== $f  overlap limit $(python tools/overlap_limit.py | head -1)"
  bash tools/seq_check.sh 3 | grep -v "ok:" 
done
