# GPU box: the whole -m gpu suite, then one bench line per workload / setting (run through gpurun)
mkdir -p gpurun_out
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_check_pytest.log 2>&1 || { tail -40 gpurun_out/gpu_check_pytest.log; exit 1; }
tail -3 gpurun_out/gpu_check_pytest.log
for args in "--frames-in-flight 1" "" "--path cells" "--workload c5" "--workload c5 --frames-in-flight 1" "--workload c2" "--workload c3sdf"; do
  echo "== $args"
  python bench.py --no-cpu-baseline --no-extra-legs $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; print(o['ms_per_frame'], round(o['value'],1), 'kernel_ms', r.get('kernel_ms'), 'spr', r.get('samples_per_ray'))
"
done
