#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/full_kernel_bench.py, three-pass form, one scene per run; prints the per-kernel statistics
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/prof3p"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for sc in "point light" "lights+mirror" "textured"; do
  tag="$(echo "$sc" | tr -c 'a-z\n' _)"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python3 "$root/tools/full_kernel_bench.py" "$sc" only3 > "$out/$tag.txt" 2> "$out/$tag.err" || echo "failed: $sc"
  echo "== $sc"; cat "$out/$tag.txt"
  f="$(find "$out/$tag" -name '*kernel_stats.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
done
