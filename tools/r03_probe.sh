# GPU box: per-rank march rate of an N-way strong-scaled frame with fused block launches (no gather)
mkdir -p gpurun_out
out=gpurun_out/r03_strong_scaling_probe.txt
: > $out
for G in 24 48 96; do
  echo "### frames per launch G=$G, 8-row strips" >> $out
  PROBE_STRIP_ROWS=8 PROBE_BLOCK_FRAMES=$G PROBE_FRAMES_IN_FLIGHT=1,2,3 python tools/strong_scaling_probe.py c3 >> $out 2>gpurun_out/r03_probe.err
done
echo "### config 4 (3840x2160), G=24" >> $out
PROBE_STRIP_ROWS=8 PROBE_BLOCK_FRAMES=24 PROBE_FRAMES_IN_FLIGHT=1,2 python tools/strong_scaling_probe.py c4 >> $out 2>>gpurun_out/r03_probe.err
cat $out
