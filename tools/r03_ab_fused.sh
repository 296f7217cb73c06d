#!/bin/bash
# A/B of the fused-launch build variants. "build" here (no GPU), "run" on the GPU box: every variant in its own process, two rounds, interleaved.
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
variants=(${VRT_AB_VARIANTS:-"base:" "norot:-DVRT_AB_NO_ROTATE" "earlypix:-DVRT_AB_EARLY_PIX" "f32:-DVRT_MAX_BLOCK_FRAMES=32" "all_old:-DVRT_AB_NO_ROTATE+-DVRT_AB_EARLY_PIX+-DVRT_MAX_BLOCK_FRAMES=32"})
mode="${1:-build}"; shift || true
if [ "$mode" = build ]; then
  for v in "${variants[@]}"; do
    name="${v%%:*}"; defs="${v#*:}"; defs="${defs//+/ }"
    VRT_BUILD_TMP="/tmp/vrtbuild_ab_$name" VRT_LIB_NAME="ab_$name.so" VRT_EXTRA_DEFS="$defs" bash "$root/volumetricraytracer_amd/csrc/build.sh" || exit 1
  done
  exit 0
fi
out="$root/gpurun_out/ab_fused"; mkdir -p "$out"; : > "$out/ab.txt"
for round in 1 2; do
  for v in "${variants[@]}"; do
    name="${v%%:*}"
    for cfg in "1 32" "3 16"; do
      set -- $cfg
      VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --frames-in-flight $1 --block-frames $2 2> "$out/$name.err" | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; print('$name round $round K=$1 G=$2: us/frame', round(o['ms_per_frame']*1e3,2), 'kernel_ms', r.get('kernel_ms'))
" >> "$out/ab.txt"
    done
  done
done
cat "$out/ab.txt"
