#!/bin/bash
# Runs on the GPU box (via gpurun): the evidence behind bench.py's `roofline` object for one workload / setting.
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel time)
#   2./3. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (they cannot share a pass on gfx950): HBM bytes
#   4. one --pmc pass of SQ counters: vector instructions, wave quad-cycles, GPU cycles
# Counters in their own runs, never with a tracing domain.  Usage: tools/r05_profile.sh <tag> [bench args...]
# Outputs under gpurun_out/prof_<tag>/; summary_<tag>.txt / counters_<tag>.json are what gets copied into profiles/.
set -uo pipefail
tag="$1"; shift
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 "$@" > "$out/bench_trace.json" 2> "$out/trace.err" || echo "trace run failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_fetch.json" 2> "$out/fetch.err" || echo "fetch run failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_write.json" 2> "$out/write.err" || echo "write run failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$out/pmc_sq" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_sq.json" 2> "$out/sq.err" || echo "sq run failed"
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc_sq2" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_sq2.json" 2> "$out/sq2.err" || echo "sq2 run failed"
# texture path: address (TA) and data-return (TD) units busy cycles, summed over the 256 CUs (two small passes: larger TA/TD sets exceed what the
# hardware collects at once, and rocprofv3 then aborts and hangs — hence the time limits)
timeout -k 10 200 rocprofv3 --pmc TD_TD_BUSY_sum TA_TA_BUSY_sum --output-format csv -d "$out/pmc_tatd" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_tatd.json" 2> "$out/tatd.err" || echo "ta/td run failed"
timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d "$out/pmc_tcp" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_tcp.json" 2> "$out/tcp.err" || echo "tcp run failed"
if [ -n "${VRT_PROFILE_KERNELS:-}" ]; then  # one summary per named kernel (the passes of the full closest hit)
  for kn in $VRT_PROFILE_KERNELS; do python3 tools/r03_summarize.py "$out" "${tag}_${kn}" "$kn" || true; done
else
  python3 tools/r03_summarize.py "$out" "$tag" || true
fi
