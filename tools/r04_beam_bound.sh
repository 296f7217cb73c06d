#!/bin/bash
# The beam pre-pass's upper bound (profiles/r04_beam_prepass_upper_bound.txt).  "build" HERE (no GPU needed): cross-compiles the A/B library
# lib/ab_tstart.so (-DVRT_AB_TSTART: records / uses every camera ray's first sampled position); "run" on the GPU box (gpurun).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if [ "${1:-build}" = build ]; then
  VRT_BUILD_TMP=/tmp/vrtbuild_ab_tstart VRT_LIB_NAME=ab_tstart.so VRT_EXTRA_DEFS="-DVRT_AB_TSTART" bash "$root/volumetricraytracer_amd/csrc/build.sh"
  exit 0
fi
for w in c3 cover; do
  VRT_LIB="$root/volumetricraytracer_amd/lib/ab_tstart.so" python3 "$root/tools/beam_upper_bound.py" $w
done
