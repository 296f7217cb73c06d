#!/bin/bash
# Builds the C-ABI library for gfx950 in-tree: volumetricraytracer_amd/lib/libvrt_hip.so
# (kept out of git by .gitignore, shipped to the GPU box by gpurun).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
tmp="${VRT_BUILD_TMP:-/tmp/vrtbuild}"
mkdir -p "$out" "$tmp"
cd "$tmp"
# VRT_LIB_NAME / VRT_EXTRA_DEFS: A/B builds of kernel variants next to the product library (tools/ab_lib_variants.sh)
name="${VRT_LIB_NAME:-libvrt_hip.so}"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wall -Wextra \
      -save-temps=obj ${VRT_EXTRA_DEFS:-} \
      -o "$out/$name" "$here/vrt_api.hip" "$here/vrt_kernels.hip"
# -save-temps=obj drops the intermediates next to the output; keep the ISA in $tmp, drop the rest
mv "$out"/vrt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s "$tmp"/ 2>/dev/null || true
rm -f "$out"/vrt_api-* "$out"/vrt_kernels-* "$out"/*.hipfb
echo "built $out/$name"
