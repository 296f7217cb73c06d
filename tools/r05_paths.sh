#!/bin/bash
# GPU box: the data paths of the march side by side on one box (two interleaved rounds): bricks (default), x-plane records (VRT_PATH_PLANES) in
# both device formats (needs tools/experiments/r05_plane_records.patch applied: the path did not meet the adoption bar), int16 bricks, cell records; on c3 (the metric), c3cover (every wave marches) and c5.  Usage: tools/r05_paths.sh [rounds]
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/r05/paths"; mkdir -p "$out"; : > "$out/paths.txt"
rounds="${1:-2}"
for round in $(seq 1 "$rounds"); do
  for a in "auto:f32" "planes:f32" "auto:texel16" "planes:texel16" "cells:texel16"; do
    path="${a%%:*}"; fmt="${a#*:}"
    for w in c3 c3cover c5; do
      timeout -k 10 200 python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --workload $w --path $path --format $fmt > "$out/$path.$fmt.$w.$round.json" 2> "$out/$path.$fmt.$w.$round.err" || echo "$a $w failed"
      python3 - "$path" "$fmt" "$w" "$round" "$out/$path.$fmt.$w.$round.json" >> "$out/paths.txt" <<'PY'
import json, sys
path, fmt, w, rnd, f = sys.argv[1:6]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"--path {path:7s} --format {fmt:8s} {w:8s} round {rnd}: {j['value']/1e3:7.2f} Grays/s  {j['ms_per_frame']*1e3:7.2f} us/frame  kernel {j['roofline']['kernel_ms']:8.4f} ms/launch")
except Exception as e:
    print(f"--path {path} --format {fmt} {w} round {rnd}: no result ({e})")
PY
    done
  done
done
cat "$out/paths.txt"
