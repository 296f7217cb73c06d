#!/bin/bash
# Kernel time and throughput of every bench workload, one and two frames in flight (runs on the GPU box).
for w in c3 c3sdf c2 c5 c4; do for k in 1 2; do python bench.py --no-cpu-baseline --workload $w --frames-in-flight $k 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('$w K=$k', 'ms/frame', o['ms_per_frame'], 'kernel_us', round(o['roofline']['kernel_ms']*1e3,1), 'Grays/s', round(o['value']/1e3,2), 'frac', o['roofline']['frac'])
"; done; done
