#!/bin/bash
# By hand on the GPU box: the default bench (no CPU baseline, 20 steps) with the product library and with every A/B build named on the
# command line (library file names under volumetricraytracer_amd/lib), interleaved twice so that a drift of the box shows up; plus a
# bit-for-bit comparison of one frame of four workloads.  Output: gpurun_out/r04/ab_<tag>.txt
#   tools/ab_bench_variants.sh <tag> libvrt_hip_a.so libvrt_hip_b.so ...
set -uo pipefail
tag="$1"; shift
out="gpurun_out/r04/ab_${tag}.txt"
mkdir -p gpurun_out/r04
: > "$out"
timeout -k 10 300 python tools/ab_frames.py /tmp/ab_product.npz >/dev/null 2>&1 || exit 1
for lib in "$@"; do
  VRT_LIB="volumetricraytracer_amd/lib/$lib" timeout -k 10 300 python tools/ab_frames.py "/tmp/ab_$lib.npz" >/dev/null 2>&1 || exit 1
  echo "== $lib against the product library, frames" >> "$out"
  python tools/ab_frames.py --compare /tmp/ab_product.npz "/tmp/ab_$lib.npz" | grep -v timing >> "$out"
done
for round in 1 2; do
  for lib in libvrt_hip.so "$@"; do
    VRT_LIB="volumetricraytracer_amd/lib/$lib" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lib round $round: %.1f Mrays/s  kernel %.4f ms/launch  samples/ray %s' % (d['value'], r.get('kernel_ms', 0), d['config'].get('samples_per_ray')))" >> "$out" || exit 1
  done
done
cat "$out"
