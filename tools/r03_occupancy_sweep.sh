#!/bin/bash
# GPU box: cap resident waves per CU with untouched dynamic LDS (VRT_AB_LDS_BYTES) — is the march limited by wave slots or by what the waves share?
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/r03_occupancy_sweep.txt"; : > "$out"
for lib in ${VRT_AB_LIBS:-nosky}; do
for lds in 0 5120 5632 6656 8192 10240 16384; do
  VRT_AB_LDS_BYTES=$lds VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$lib.so" python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('$lib lds $lds B/wave -> max', min(32, 163840//max($lds,1)) if $lds else 32, 'waves/CU: us/frame', round(o['ms_per_frame']*1e3,2), 'Grays/s', round(o['value']/1e3,2))
" >> "$out"
done
done
cat "$out"
