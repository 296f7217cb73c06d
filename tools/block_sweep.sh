#!/bin/bash
# GPU box: bench.py (frames issued as vrt_render_block blocks) at several block sizes, with HIP's default 4 hardware queues and with 8
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/block_sweep.txt"; mkdir -p "$root/gpurun_out"; : > "$out"
for q in default 8; do
  for K in 1 2 3 4 6 8; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 10 --frames-in-flight $K "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('hw queues $q K=$K: %.1f us/frame %.2f Grays/s (kernel %.1f us)' % (o['ms_per_frame'] * 1e3, o['value'] / 1e3, r['kernel_ms'] * 1e3))
" >> "$out"
  done
done
cat "$out"
