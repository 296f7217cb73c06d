#!/bin/bash
# GPU box: libraries named on the command line (volumetricraytracer_amd/lib/<name>.so) x k_relax values, interleaved, two rounds.
# Usage: tools/ab_lib_relax.sh "ab_prerelax:1.0 libvrt_hip:1.0 libvrt_hip:1.7"
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/ab_lib_relax.txt"; mkdir -p "$root/gpurun_out"; : > "$out"
for round in 1 2; do
  for spec in $1; do
    lib="${spec%%:*}"; k="${spec#*:}"
    for K in 1 3; do
      VRT_LIB="$root/volumetricraytracer_amd/lib/$lib.so" python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 10 --k-relax $k --frames-in-flight $K 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('round $round $lib k_relax $k K=$K: %.1f us/frame %.2f Grays/s (kernel %.1f us) samples/ray %.2f' % (o['ms_per_frame'] * 1e3, o['value'] / 1e3, r['kernel_ms'] * 1e3, o['config']['samples_per_ray']))
" >> "$out"
    done
  done
done
cat "$out"
