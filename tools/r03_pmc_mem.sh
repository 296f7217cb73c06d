#!/bin/bash
# GPU box: memory-side counters of the fused march launch (TA / TCP / TCC), counters only.  Usage: tools/r03_pmc_mem.sh [bench args]
set -uo pipefail
out="gpurun_out/pmc_r03_mem"; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for set in \
  "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TD_TD_BUSY_sum TA_TA_BUSY_sum" ; do
  i=$((i+1))
  echo "pass $i: $set" >> "$out/progress.log"
  # (a counter set the hardware cannot collect makes rocprofv3 abort and then hang: every pass has its own time limit)
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 "$@" > "$out/bench_p$i.json" 2> "$out/p$i.err" || echo "pass $i failed: $(grep -m1 'error code' $out/p$i.err)" | tee -a "$out/progress.log"
done
python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float)); grid = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r.get("Kernel_Name", ""):
            key = (f, r["Dispatch_Id"])
            tot[r["Counter_Name"]][key] += float(r["Counter_Value"]); grid[key] = int(r.get("Grid_Size") or 0)
gmax = max(grid.values())
lines = []
for k, v in sorted(tot.items()):
    vals = [x for kk, x in v.items() if grid[kk] == gmax]
    lines.append(f"{k:40s} {sum(vals) / max(len(vals), 1):18.1f}   (mean per full 48-frame launch)")
open(os.path.join(out, "pmc_mem.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
