#!/bin/bash
# Round-2 evidence run on the GPU box: everything profiles/r02_* is made from.  Usage: tools/r02_profiles.sh
set -uo pipefail
out=gpurun_out/r02p; mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tools/profile_bench.sh r02 > $out/profile.log 2>&1
tools/pmc_sq.sh r02 --no-extra-legs > $out/pmc.log 2>&1
python3 tools/timeline.py c3 > $out/timeline_c3.txt 2>&1
for k in 1 2 3 4; do python3 bench.py --no-cpu-baseline --no-extra-legs --frames-in-flight $k > $out/k$k.json 2>/dev/null; done
for cfg in "f32:--format f32" "texel16_bricks:--format texel16" "texel16_cells:--format texel16 --path cells" "f32_dense:--path dense" "f32_lds:--path lds"; do
  name="${cfg%%:*}"; args="${cfg#*:}"
  python3 bench.py --no-cpu-baseline $args > $out/fmt_$name.json 2>/dev/null
done
for w in c2 c3sdf c5; do python3 bench.py --no-cpu-baseline --no-extra-legs --workload $w > $out/wl_$w.json 2>/dev/null; done
(cd tools/microbench && { [ -x gather16 ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gather16 gather16.hip; } && ./gather16 > ../../$out/gather16.txt 2>&1)
bash tools/relax_sweep.sh > $out/relax_sweep.log 2>&1; cp gpurun_out/relax_sweep.txt $out/relax_sweep.txt
# libraries of earlier commits built next to the product one (volumetricraytracer_amd/lib/ab_*.so), when present
libs=""; for l in ab_prerelax:1.0 ab_prev:1.7 ab_head:1.7 libvrt_hip:1.7; do [ -f volumetricraytracer_amd/lib/${l%%:*}.so ] && libs="$libs $l"; done
bash tools/ab_lib_relax.sh "$libs" > $out/ab_lib_relax.log 2>&1; cp gpurun_out/ab_lib_relax.txt $out/ab_lib_relax.txt
python3 - <<'PY'
import json, glob, os
out = "gpurun_out/r02p"
def j(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return None
lines = ["frames in flight sweep (config 3, f32 bricks): K  ms/frame  Grays/s  kernel_ms(events)"]
for k in (1, 2, 3, 4):
    r = j(f"{out}/k{k}.json")
    if r: lines.append(f"  {k}  {r['ms_per_frame']:.4f}  {r['value']/1e3:6.2f}  {r['roofline']['kernel_ms']:.4f}")
lines.append("")
lines.append("device formats / data paths (config 3, K=3): name  ms/frame  Grays/s | K=1 ms/frame | 4K ms/frame Grays/s")
for p in sorted(glob.glob(f"{out}/fmt_*.json")):
    r = j(p)
    if r:
        lines.append(f"  {os.path.basename(p)[4:-5]:16s} {r['ms_per_frame']:.4f}  {r['value']/1e3:6.2f} | {r['latency']['ms_per_frame']:.4f} | {r['config4']['ms_per_frame']:.4f} {r['config4']['value']/1e3:6.2f}")
lines.append("")
lines.append("other workloads (K=3): name  ms/frame  Grays/s  kernel_ms  rays/frame  samples/ray")
for p in sorted(glob.glob(f"{out}/wl_*.json")):
    r = j(p)
    if r: lines.append(f"  {os.path.basename(p)[3:-5]:6s} {r['ms_per_frame']:.4f}  {r['value']/1e3:6.2f}  {r['roofline']['kernel_ms']:.4f}  {r['config']['rays_per_frame']}  {r['config']['samples_per_ray']}")
open(f"{out}/sweeps.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
