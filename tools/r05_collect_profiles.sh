#!/bin/bash
# HERE (after the gpurun calls of tools/r05_profile.sh / r05_profiles_all.sh have merged gpurun_out/prof_r05_*): copies the summaries that
# are to be judged into profiles/ under their round-5 names; counters of the default bench command become profiles/counters_latest.json.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root"
for d in gpurun_out/prof_r05_*; do
  tag="${d#gpurun_out/prof_}"; w="${tag#r05_}"
  for s in "$d"/summary_"$tag"*.txt; do
    [ -f "$s" ] || continue
    b="$(basename "$s" .txt)"; b="${b#summary_}"
    cp "$s" "profiles/${b}_rocprofv3_kernel_stats_and_counters.txt"
  done
  for c in "$d"/counters_"$tag"*.json; do
    [ -f "$c" ] || continue
    b="$(basename "$c" .json)"; b="${b#counters_r05_}"
    cp "$c" "profiles/r05_counters_${b}.json"
  done
done
cp profiles/r05_counters_c3.json profiles/counters_latest.json
ks="$(find gpurun_out/prof_r05_c3/trace -name '*kernel_stats.csv' | head -1)"
[ -n "$ks" ] && cp "$ks" profiles/r05_kernel_stats.csv
ls -la profiles | grep r05_
