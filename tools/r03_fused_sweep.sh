# GPU box: fused-launch sweep — frames per launch (G) x streams (K), against per-frame launches (round 2's behaviour)
mkdir -p gpurun_out
out=gpurun_out/r03_fused_sweep.txt
: > $out
run() {
  echo "== $*" >> $out
  python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 "$@" 2>>gpurun_out/r03_fused_sweep.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; print('ms/frame', o['ms_per_frame'], 'Mrays/s', round(o['value'],1), 'kernel_ms', r.get('kernel_ms'), 'frames/launch', r.get('frames_per_launch'), 'frac', r.get('frac'))
" >> $out
}
for K in 1 2 3 4; do
  run --frames-in-flight $K --block-frames 2 --per-frame-launches
done
for G in 4 8 16 32; do
  for K in 1 2 3; do
    run --frames-in-flight $K --block-frames $G
  done
done
run --frames-in-flight 1 --block-frames 64
run --frames-in-flight 2 --block-frames 64
cat $out
