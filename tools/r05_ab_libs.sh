#!/bin/bash
# GPU box: two builds of the library side by side on one box (interleaved rounds): volumetricraytracer_amd/lib/ab_base.so (built from an
# earlier commit's kernels by hand) against the product library, on c3 (the metric), c3cover (every wave marches), c5 (BVH) and c3dropin.
# VRT_AB_LIBS="a.so b.so ..." names other libraries under volumetricraytracer_amd/lib/ to put side by side.
# Usage: tools/r05_ab_libs.sh [rounds] [workloads...]
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/r05/ab_libs"; mkdir -p "$out"; : > "$out/ab.txt"
rounds="${1:-2}"; shift || true
workloads=("$@"); [ ${#workloads[@]} -eq 0 ] && workloads=(c3 c3cover c5 c3dropin)
for round in $(seq 1 "$rounds"); do
  for lib in ${VRT_AB_LIBS:-ab_base.so libvrt_hip.so}; do
    for w in "${workloads[@]}"; do
      VRT_LIB="$root/volumetricraytracer_amd/lib/$lib" timeout -k 10 200 python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --workload $w > "$out/$lib.$w.$round.json" 2> "$out/$lib.$w.$round.err" || echo "$lib $w failed"
      python3 - "$lib" "$w" "$round" "$out/$lib.$w.$round.json" >> "$out/ab.txt" <<'PY'
import json, sys
lib, w, rnd, f = sys.argv[1:5]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{lib:16s} {w:8s} round {rnd}: {j['value']/1e3:7.2f} Grays/s  {j['ms_per_frame']*1e3:7.2f} us/frame  kernel {j['roofline']['kernel_ms']:8.4f} ms/launch")
except Exception as e:
    print(f"{lib} {w} round {rnd}: no result ({e})")
PY
    done
  done
done
cat "$out/ab.txt"
