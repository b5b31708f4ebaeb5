#!/bin/bash
# usage: bash tools/build_variants.sh "_suffix -DFLAG [-DFLAG2]" ...   builds codesign-kernels_amd/libmpdata_hip<suffix>.so
# (experiment builds of the library for the interleaved A/B scripts: tools/ab_uw.sh, ab_t25.sh, ab_wm.sh)
cd "$(dirname "$0")/../codesign-kernels_amd/csrc" || exit 1
for v in "$@"; do
  set -- $v
  sfx=$1; shift
  make -j4 SUFFIX=$sfx EXTRA="$*" ../libmpdata_hip$sfx.so 2>&1 | grep -v "^hipcc\|resource usage" | tail -3
done
ls -la ../libmpdata_hip*.so
