#!/bin/bash
# GPU box: tools/full_kernel_bench.py once per library variant built by tools/r03_ab_fused.sh build (VRT_AB_VARIANTS="name:defs ...")
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
variants=(${VRT_AB_VARIANTS})
out="$root/gpurun_out/ab_full"; mkdir -p "$out"; : > "$out/ab.txt"
for round in 1 2; do
  for v in "${variants[@]}"; do
    name="${v%%:*}"
    echo "== $name round $round" >> "$out/ab.txt"
    VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" timeout -k 10 300 python3 "$root/tools/full_kernel_bench.py" >> "$out/ab.txt" 2> "$out/$name.err" || echo "FAILED $name" >> "$out/ab.txt"
  done
done
cat "$out/ab.txt"
