mkdir -p gpurun_out
out=gpurun_out/r03_fused_sweep2.txt
: > $out
run() {
  echo "== $*" >> $out
  python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 "$@" 2>>gpurun_out/r03_fused_sweep2.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; print('ms/frame', o['ms_per_frame'], 'Mrays/s', round(o['value'],1), 'kernel_ms', r.get('kernel_ms'), 'frames/launch', r.get('frames_per_launch'), 'frac', r.get('frac'))
" >> $out
}
run --frames-in-flight 3 --block-frames 2 --per-frame-launches
run --frames-in-flight 3 --block-frames 2 --per-frame-launches --frames-per-step 96
for K in 1 2 3; do
  run --frames-in-flight $K --block-frames 16
  run --frames-in-flight $K --block-frames 32
  run --frames-in-flight $K --block-frames 48 --frames-per-step 96
done
cat $out
