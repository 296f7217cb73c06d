#!/bin/bash
# GPU box: the default bench line of every workload (no extra legs, no CPU baseline), one after the other
set -uo pipefail
for w in c3 c5 c4 c3light; do
  python3 bench.py --no-cpu-baseline --no-extra-legs --workload $w --steps 10 --warmup 3 2> gpurun_out/quick_$w.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('$w', 'us/frame', round(o['ms_per_frame']*1e3,2), 'Grays/s', round(o['value']/1e3,2), 'kernel_ms', o['roofline'].get('kernel_ms'))
"
done
