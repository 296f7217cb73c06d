#!/bin/bash
# Runs on the GPU box: SQ / cache PMC passes over bench.py (counters only, no tracing domains),
# summarised per march_kernel dispatch.  Usage: tools/pmc_sq.sh <tag> [bench args...]
set -uo pipefail
tag="$1"; shift
out="gpurun_out/pmc_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
  "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F32" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > "$out/bench_p$i.json" 2> "$out/p$i.err" || echo "pass $i failed: $(tail -2 $out/p$i.err)"
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
tot = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_kernel" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
res = {k: sum(v.values()) / len(v) for k, v in tot.items()}
lines = [f"{k:32s} {v:18.1f}" for k, v in sorted(res.items())]
w = res.get("SQ_WAVES", 0)
if w:
    lines.append(f"-- per wave: VALU {res.get('SQ_INSTS_VALU',0)/w:.0f}  SALU {res.get('SQ_INSTS_SALU',0)/w:.0f}  VMEM_RD {res.get('SQ_INSTS_VMEM_RD',0)/w:.0f}  SMEM {res.get('SQ_INSTS_SMEM',0)/w:.0f}  TRANS {res.get('SQ_INSTS_VALU_TRANS_F32',0)/w:.0f}  BRANCH {res.get('SQ_INSTS_BRANCH',0)/w:.0f}")
if res.get("SQ_ACTIVE_INST_VALU") and res.get("SQ_THREAD_CYCLES_VALU"):
    lines.append(f"-- VALU lane utilisation: {res['SQ_THREAD_CYCLES_VALU'] / (res['SQ_ACTIVE_INST_VALU'] * 64) :.3f} (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64))")
if res.get("TCC_HIT_sum") is not None and res.get("TCC_MISS_sum") is not None and (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]) > 0:
    lines.append(f"-- L2 hit rate: {res['TCC_HIT_sum'] / (res['TCC_HIT_sum'] + res['TCC_MISS_sum']):.3f}")
open(os.path.join(out, f"pmc_{tag}.txt"), "w").write("\n".join(lines) + "\n")
json.dump(res, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1)
print("\n".join(lines))
PY
