#!/bin/bash
# GPU box: interleaved A/B of library variants built by tools/r03_ab_fused.sh build; default bench settings (K=1, 48 frames per launch)
# Usage: VRT_AB_VARIANTS="a: b:-DX" tools/r03_ab_run.sh [extra bench args]
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
variants=(${VRT_AB_VARIANTS})
out="$root/gpurun_out/ab_run"; mkdir -p "$out"; : > "$out/ab.txt"
for round in 1 2 3; do
  for v in "${variants[@]}"; do
    name="${v%%:*}"
    VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 "$@" 2> "$out/$name.err" | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; print('$name round $round: us/frame', round(o['ms_per_frame']*1e3,2), 'Grays/s', round(o['value']/1e3,2), 'kernel_ms', r.get('kernel_ms'))
" >> "$out/ab.txt"
  done
done
cat "$out/ab.txt"
