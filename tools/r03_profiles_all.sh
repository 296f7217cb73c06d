#!/bin/bash
# GPU box: the profile set of round 3 — config 3 (the metric), config 5 (BVH kernel), config 4 (3840x2160) and config 3 + one point light
# (march_kernel_full): rocprofv3 kernel stats + the PMC passes each (tools/r03_profile.sh)
set -uo pipefail
for w in c3 c5 c4 c3light; do
  # (config 3 + one point light runs the full closest hit in passes: one summary per pass kernel)
  if [ $w = c3light ]; then export VRT_PROFILE_KERNELS="primary_pass_kernel light_pass_kernel"; else unset VRT_PROFILE_KERNELS; fi
  bash tools/r03_profile.sh r03_$w --workload $w --steps 10 --warmup 3 > gpurun_out/r03_profile_$w.log 2>&1
  tail -12 gpurun_out/r03_profile_$w.log
done
