#!/bin/bash
# Round 5, VERDICT r4 item 3: result-neutral PLACEMENT variants against the 19-cache-lines-per-load ratio, each as an A/B on one box with
# the counter next to the time.  "build" (here, no GPU) cross-compiles every variant into volumetricraytracer_amd/lib/ab_<name>.so from
# vrt_kernels.hip + tools/experiments/r05_placement.patch's switches; "run" (GPU box) benches each variant on c3, c3cover (every wave
# marches) and c5, two interleaved rounds, then takes two small --pmc passes per variant on c3 and c3cover
# (TCP_TOTAL_CACHE_ACCESSES_sum / SQ_INSTS_VMEM_RD = cache lines per wave-level load; TD_TD_BUSY_sum, GRBM_GUI_ACTIVE = TD busy).
# Usage: tools/r05_ab_placement.sh build | run
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
variants=(${VRT_AB_VARIANTS:-"base:" "morton_lanes:-DVRT_AB_LANE_MAP=1" "wave16x4:-DVRT_AB_LANE_MAP=2" "wave4x16:-DVRT_AB_LANE_MAP=3" "morton_pool:-DVRT_AB_MORTON_POOL" "morton_lanes_pool:-DVRT_AB_LANE_MAP=1+-DVRT_AB_MORTON_POOL"})
mode="${1:-build}"; shift || true
if [ "$mode" = build ]; then
  for v in "${variants[@]}"; do
    name="${v%%:*}"; defs="${v#*:}"; defs="${defs//+/ }"
    VRT_BUILD_TMP="/tmp/vrtbuild_ab_$name" VRT_LIB_NAME="ab_$name.so" VRT_EXTRA_DEFS="$defs" bash "$root/volumetricraytracer_amd/csrc/build.sh" || exit 1
  done
  exit 0
fi
out="$root/gpurun_out/r05/ab_placement"; mkdir -p "$out"; : > "$out/ab.txt"
export TMPDIR=/tmp
# placement only: every variant must render the very frames of the base build (sha256 of a 1080p config-3 frame and of a config-5 frame)
for v in "${variants[@]}"; do
  name="${v%%:*}"
  VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" python3 - "$name" >> "$out/ab.txt" <<'PY'
import hashlib, sys
sys.path.insert(0, ".")
import volumetricraytracer_amd as v
from volumetricraytracer_amd import workloads as scenes
out = []
with v.VHipRenderer() as r:
    for sc in (scenes.bench_config3(), scenes.config5_instances(6, 32)):
        p = v.default_params(1920, 1080, scenes.min_cell(sc), 255, shadow=True)
        r.SetSceneToRender(sc); r.ResizeRenderOutput(1920, 1080); r.params_override = p
        out.append(hashlib.sha256(r.Render().tobytes()).hexdigest()[:16])
print(f"{sys.argv[1]:20s} frames {out}")
PY
done
for round in 1 2; do
  for v in "${variants[@]}"; do
    name="${v%%:*}"
    for w in c3 c3cover c5; do
      VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" timeout -k 10 200 python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --workload $w > "$out/$name.$w.$round.json" 2> "$out/$name.$w.$round.err" || echo "$name $w failed"
      python3 - "$name" "$w" "$round" "$out/$name.$w.$round.json" >> "$out/ab.txt" <<'PY'
import json, sys
name, w, rnd, path = sys.argv[1:5]
try:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"{name:20s} {w:8s} round {rnd}: {j['value']/1e3:7.2f} Grays/s  {j['ms_per_frame']*1e3:7.2f} us/frame  kernel {j['roofline']['kernel_ms']:8.4f} ms/launch  samples/ray {j['config']['samples_per_ray']}")
except Exception as e:
    print(f"{name:20s} {w:8s} round {rnd}: no result ({e})")
PY
    done
  done
done
echo "== counters (mean per full launch) ==" >> "$out/ab.txt"
for v in "${variants[@]}"; do
  name="${v%%:*}"
  for w in c3 c3cover; do
    d="$out/pmc_${name}_$w"
    VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$d/tcp" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs --workload $w > "$d.tcp.json" 2> "$d.tcp.err" || echo "tcp pass failed: $name $w"
    VRT_LIB="$root/volumetricraytracer_amd/lib/ab_$name.so" timeout -k 10 200 rocprofv3 --pmc TD_TD_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d "$d/tatd" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs --workload $w > "$d.tatd.json" 2> "$d.tatd.err" || echo "tatd pass failed: $name $w"
    python3 - "$name" "$w" "$d" >> "$out/ab.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
name, w, d = sys.argv[1:4]
def counters(sub):
    per, grid = defaultdict(lambda: defaultdict(float)), {}
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "march_kernel" not in r.get("Kernel_Name", ""):
                continue
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            grid[r["Dispatch_Id"]] = int(r.get("Grid_Size") or 0)
    if not grid:
        return {}
    gmax = max(grid.values())
    keep = [k for k, g in grid.items() if g == gmax]
    return {c: sum(v[k] for k in keep if k in v) / max(len([k for k in keep if k in v]), 1) for c, v in per.items()}
a, b = counters("tcp"), counters("tatd")
try:
    cyc_a, cyc_b = a["GRBM_GUI_ACTIVE"] / 8.0, b["GRBM_GUI_ACTIVE"] / 8.0
    print(f"{name:20s} {w:8s} lines per wave-level load {a['TCP_TOTAL_CACHE_ACCESSES_sum'] / a['SQ_INSTS_VMEM_RD']:6.2f}  L1 line look-ups per CU and cycle {a['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256.0 / cyc_a:5.3f}  "
          f"VMEM_RD per launch {a['SQ_INSTS_VMEM_RD'] / 1e6:8.2f} M  TD busy {b['TD_TD_BUSY_sum'] / 256.0 / cyc_b:5.3f}  TA busy {b['TA_TA_BUSY_sum'] / 256.0 / cyc_b:5.3f}")
except Exception as e:
    print(f"{name:20s} {w:8s} counters incomplete ({e}): {sorted(a)} {sorted(b)}")
PY
  done
done
# N1 (the north-star's LDS staging) re-timed on this build: the hybrid LDS kernel against the default path, same box
for p in auto lds; do
  timeout -k 10 200 python3 "$root/bench.py" --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --path $p > "$out/path_$p.json" 2> "$out/path_$p.err" || echo "path $p failed"
  python3 - "$p" "$out/path_$p.json" >> "$out/ab.txt" <<'PY'
import json, sys
p, path = sys.argv[1:3]
try:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"--path {p:5s} (product library) c3: {j['value']/1e3:7.2f} Grays/s  kernel {j['roofline']['kernel_ms']:8.4f} ms/launch")
except Exception as e:
    print(f"--path {p}: no result ({e})")
PY
done
cat "$out/ab.txt"
