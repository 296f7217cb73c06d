#!/bin/bash
# A/B of compile-time kernel variants.  Run HERE (no GPU) with "build" to cross-compile every variant into
# volumetricraytracer_amd/lib/ab_<name>.so (they travel to the GPU box with the snapshot), and on the GPU box with "run"
# to bench each one in its own process (K = 1 latency and K = 2 throughput), two rounds, interleaved.
# Usage: tools/ab_lib_variants.sh build | run [bench args...]
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
variants=(${VRT_AB_VARIANTS:-"base:" "incell:-DVRT_AB_INCELL" "spec:-DVRT_AB_SPEC" "no_brick_cache:-DVRT_AB_NO_BRICK_CACHE" "quad_blocks:-DVRT_AB_QUAD_BLOCKS"})
mode="${1:-build}"; shift || true
if [ "$mode" = build ]; then
  for v in "${variants[@]}"; do
    name="${v%%:*}"; defs="${v#*:}"
    VRT_BUILD_TMP="/tmp/vrtbuild_ab_$name" VRT_LIB_NAME="ab_$name.so" VRT_EXTRA_DEFS="$defs" bash "$root/volumetricraytracer_amd/csrc/build.sh" || exit 1
  done
  exit 0
fi
out="$root/gpurun_out/ab"; mkdir -p "$out"; : > "$out/ab.txt"
for round in 1 2; do
  for v in "${variants[@]}"; do
    name="${v%%:*}"
    VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" python3 "$root/bench.py" --no-cpu-baseline --steps 10 "$@" > "$out/$name.$round.json" 2> "$out/$name.$round.err" || echo "$name failed"
    python3 - "$name" "$round" "$out/$name.$round.json" >> "$out/ab.txt" <<'PY'
import json, sys
name, rnd, path = sys.argv[1:4]
try:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"{name:16s} round {rnd}: K=2 {j['ms_per_frame']*1e3:7.1f} us/frame {j['value']/1e3:6.2f} Grays/s (kernel {j['roofline']['kernel_ms']*1e3:6.1f} us) | K=1 {j['latency']['ms_per_frame']*1e3:7.1f} us/frame (kernel {j['latency']['kernel_ms']*1e3:6.1f} us) | 4K {j['config4']['ms_per_frame']*1e3:7.1f} us/frame")
except Exception as e:
    print(f"{name:16s} round {rnd}: no result ({e})")
PY
  done
done
cat "$out/ab.txt"
