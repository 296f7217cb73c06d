#!/usr/bin/env python3
"""Condenses a tools/r03_profile.sh output directory into summary_<tag>.txt and counters_<tag>.json (copied into profiles/; the
latter, as profiles/counters_latest.json, is what bench.py attaches to its `roofline` object when its own settings and kernel
source hash equal the recorded key).  Only the FULL march launches count (largest grid of the run: the blocks of frames of the
timed region; the single-frame launches that count rays are left out).  HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE
and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (calibrated for this kernel's
tap patterns: profiles/r01_fetch_size_calibration.txt, r02_fetch_size_calibration_int16.txt) — raw and x2-corrected are recorded.
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "march_kernel"  # substring of the kernel's name (the passes of the full closest hit: primary_pass_kernel / light_pass_kernel)
summary = {"tag": tag}
lines = []


def grid_total(r):
    """Work-items of a dispatch: Grid_Size (counter_collection.csv) or Grid_Size_X x _Y x _Z (kernel_trace.csv)."""
    try:
        if r.get("Grid_Size"):
            return int(r["Grid_Size"])
        return int(r.get("Grid_Size_X") or 0) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
    except ValueError:
        return 0


stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    for r in rows[:10]:
        lines.append("  ".join(f"{k}={r[k]}" for k in r))
trace = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if KERNEL in r.get("Kernel_Name", "")]
    if rows:
        gmax = max(grid_total(r) for r in rows)
        full = [r for r in rows if grid_total(r) == gmax]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in full]
        summary["march_kernel_full_launches"] = {"calls": len(full), "avg_ns": sum(durs) / len(durs), "min_ns": min(durs), "max_ns": max(durs), "grid_size": gmax}
        r = full[-1]
        summary["march_kernel_dispatch"] = {k: r.get(k) for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X")}
        lines.append(f"== march kernel, full launches only (grid {gmax}): {len(full)} calls, avg {sum(durs) / len(durs) / 1e3:.1f} us, min {min(durs) / 1e3:.1f}, max {max(durs) / 1e3:.1f} ==")
        lines.append(json.dumps(summary["march_kernel_dispatch"]))


def counters(dirname):
    """{counter: mean per full launch} and the mean duration (ns) of those launches in that pass."""
    files = glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(float))
    grid, span = {}, {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if KERNEL not in r.get("Kernel_Name", ""):
                continue
            d = r["Dispatch_Id"]
            per[r["Counter_Name"]][d] += float(r["Counter_Value"])
            grid[d] = grid_total(r)
            try:
                span[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    if not grid:
        return {}, None
    gmax = max(grid.values())
    keep = [d for d, g in grid.items() if g == gmax]
    res = {k: sum(v[d] for d in keep if d in v) / max(len([d for d in keep if d in v]), 1) for k, v in per.items()}
    ns = [span[d] for d in keep if d in span]
    return res, (sum(ns) / len(ns) if ns else None)


c = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_tatd", "pmc_tcp"):
    res, ns = counters(d)
    c.update(res)
    if d == "pmc_sq" and ns:
        c["_sq_pass_launch_ns"] = ns
for k in sorted(c):
    if not k.startswith("_"):
        lines.append(f"{k:28s} {c[k]:18.1f}   (mean per full launch)")

bench = {}
for b in ("bench_trace.json", "bench_fetch.json", "bench_write.json", "bench_sq.json"):
    p = os.path.join(out, b)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        try:
            bench[b] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            pass
summary["bench_lines"] = {k: {"value": v.get("value"), "ms_per_frame": v.get("ms_per_frame"), "kernel_ms": v.get("roofline", {}).get("kernel_ms"),
                              "frames_per_launch": v.get("roofline", {}).get("frames_per_launch")} for k, v in bench.items()}
lines.append("== bench lines of the passes (value Mrays/s, ms/frame, event-timed kernel ms per launch) ==")
for k, v in summary["bench_lines"].items():
    lines.append(f"{k}: {v}")

res = {"tag": tag}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rd, wr = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    res.update({"FETCH_SIZE_KiB_per_launch": c["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": c["WRITE_SIZE"],
                "hbm_bytes_per_launch_raw": rd + wr, "hbm_bytes_per_launch": 2 * rd + wr})
    lines.append(f"== HBM traffic per launch: read {rd / 1e6:.2f} MB raw ({2 * rd / 1e6:.2f} MB x2-corrected), write {wr / 1e6:.2f} MB ==")
for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU",
          "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "TD_TD_BUSY_sum", "TA_TA_BUSY_sum",
          "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_ACCESSES_sum"):
    if k in c:
        res[k] = c[k]
if c.get("GRBM_GUI_ACTIVE"):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # per XCD = the launch's duration in GPU cycles
    res["gpu_cycles_per_launch"] = cyc
    if c.get("_sq_pass_launch_ns"):
        res["clock_ghz"] = cyc / c["_sq_pass_launch_ns"]
        lines.append(f"== clock during the SQ pass: {res['clock_ghz']:.3f} GHz ({cyc:.0f} cycles in {c['_sq_pass_launch_ns'] / 1e3:.1f} us) ==")
    if c.get("SQ_WAVE_CYCLES"):
        res["occupancy_mean_waves_per_cu"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / (cyc * 256.0), 2)
        lines.append(f"== mean occupancy: {res['occupancy_mean_waves_per_cu']} of 32 waves per CU (SQ_WAVE_CYCLES x 4 / (GPU cycles x 256 CUs)) ==")
    if c.get("TD_TD_BUSY_sum"):
        res["td_busy_frac"] = round(c["TD_TD_BUSY_sum"] / 256.0 / cyc, 4)
        res["ta_busy_frac"] = round(c.get("TA_TA_BUSY_sum", 0.0) / 256.0 / cyc, 4)
        lines.append(f"== texture path busy (sum over the CUs / 256 / GPU cycles of the SQ pass): data return TD {res['td_busy_frac']}, addresser TA {res['ta_busy_frac']} ==")
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        res["l1_cache_line_accesses_per_cu_cycle"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc, 3)
        lines.append(f"== L1 (TCP): {res['l1_cache_line_accesses_per_cu_cycle']} cache-line accesses per CU and cycle; {c.get('TCP_TCC_READ_REQ_sum', 0) / max(c['TCP_TOTAL_CACHE_ACCESSES_sum'], 1):.3f} of them go on to L2 ==")
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("SQ_INSTS_VMEM_RD"):
        res["lines_per_vmem_instr"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"], 2)
        lines.append(f"== cache lines per wave-level vector-memory read instruction: {res['lines_per_vmem_instr']} (TCP_TOTAL_CACHE_ACCESSES_sum / SQ_INSTS_VMEM_RD; ~4 if the 64 lanes of a load read one brick) ==")
    if c.get("SQ_INSTS_VALU"):
        res["valu_issue_frac_in_sq_pass"] = round(c["SQ_INSTS_VALU"] * 2.0 / 1024.0 / cyc, 4)
        lines.append(f"== vector-instruction issue: {c['SQ_INSTS_VALU']:.0f} wave-instructions x 2 cycles / 1024 SIMDs / {cyc:.0f} cycles = {res['valu_issue_frac_in_sq_pass']} (in the SQ pass itself) ==")
if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
    res["valu_lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 3)
    lines.append(f"== VALU lane utilisation: {res['valu_lane_utilisation']} ==")
if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
    lines.append(f"== per wave: VALU {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.0f}  SALU {c.get('SQ_INSTS_SALU', 0) / c['SQ_WAVES']:.0f}  VMEM_RD {c.get('SQ_INSTS_VMEM_RD', 0) / c['SQ_WAVES']:.0f} ==")
key = None
for b in ("bench_sq.json", "bench_fetch.json", "bench_trace.json"):
    key = key or bench.get(b, {}).get("roofline", {}).get("traffic_key")
if key:
    res["key"] = key
res["note"] = ("separate rocprofv3 --pmc passes over `bench.py --steps 3 --warmup 1 --no-extra-legs` (tools/r03_profile.sh), mean over the full "
               "march launches; KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950); SQ quad-cycle counters x 4")
json.dump(res, open(os.path.join(out, f"counters_{tag}.json"), "w"), indent=1)
open(os.path.join(out, f"summary_{tag}.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, f"summary_{tag}.json"), "w"), indent=1)
print("\n".join(lines))
