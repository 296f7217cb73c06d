#!/bin/bash
# GPU box: the profile set of round 5 besides config 3 (tools/r05_profile.sh r05_c3): the adaptor's defaults (the lean kernel's REF instantiation), every
# wave marching, config 5 (BVH kernel), config 4 (3840x2160) and config 3 + one point light (the full closest hit in passes): rocprofv3 kernel stats + the PMC passes each
set -uo pipefail
for w in ${VRT_PROFILE_WORKLOADS:-c3dropin c3cover c5 c4 c3light}; do
  if [ $w = c3light ]; then export VRT_PROFILE_KERNELS="primary_pass_kernel light_pass_kernel"; else unset VRT_PROFILE_KERNELS; fi
  bash tools/r05_profile.sh r05_$w --workload $w > gpurun_out/r05_profile_$w.log 2>&1
  tail -14 gpurun_out/r05_profile_$w.log
done
