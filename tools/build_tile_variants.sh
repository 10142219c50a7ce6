#!/usr/bin/env bash
# Experimental builds that force one NT/TN tile shape for every product (tools/build/tile{0,1,2}/libmmsum_hip.so: 256x256, 256x128, 128x128),
# to measure the tile rule (choose_tile) on small-batch shapes.  Not shipped.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
for t in 1 2; do
  D="$R/tools/build/tile$t"; rm -rf "$D"; mkdir -p "$D"
  cp "$R"/multimodalsum_amd/csrc/*.hip "$R"/multimodalsum_amd/csrc/*.h "$R"/multimodalsum_amd/csrc/*.inc "$R"/multimodalsum_amd/csrc/Makefile "$R"/multimodalsum_amd/csrc/check_resources.py "$D"/
  sed -i "s|inline int choose_tile(const GemmArgs& a) {|inline int choose_tile(const GemmArgs\& a) {\n    if (a.M <= 32768) return $t;|" "$D"/gemm_fast.hip
  sed -i "s|../../include/mmsum_hip.h|$R/include/mmsum_hip.h|g" "$D"/Makefile "$D"/*.hip "$D"/*.h
  (cd "$D" && make -j4 > build.log 2>&1 && echo "built $D")
done
