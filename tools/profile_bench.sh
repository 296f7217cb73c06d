#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summary of bench.py, then two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) for HBM traffic.
# Usage: tools/profile_bench.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/.
set -uo pipefail
tag="$1"; shift
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline --no-extra-legs "$@" > "$out/bench_trace.json" 2> "$out/trace.err" || echo "trace run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_fetch.json" 2> "$out/fetch.err" || echo "fetch run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-extra-legs "$@" > "$out/bench_write.json" 2> "$out/write.err" || echo "write run failed"
find "$out" -name "*.csv" | head -30
python3 tools/summarize_profile.py "$out" "$tag" || true
