#!/bin/bash
# GPU box: the over-relaxation factor of the sphere trace (vrt_params.k_relax) against frame time, interleaved in one call,
# two rounds: 1 = plain sphere tracing.  Output: gpurun_out/relax_sweep.txt
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/relax_sweep.txt"; mkdir -p "$root/gpurun_out"; : > "$out"
for round in 1 2; do
  for k in 1.0 1.4 1.7 2.0; do
    for K in 1 3; do
      python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 10 --k-relax $k --frames-in-flight $K "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('round $round k_relax $k K=$K: %.1f us/frame %.2f Grays/s (kernel %.1f us) samples/ray %.2f' % (o['ms_per_frame'] * 1e3, o['value'] / 1e3, r['kernel_ms'] * 1e3, o['config']['samples_per_ray']))
" >> "$out"
    done
  done
done
cat "$out"
