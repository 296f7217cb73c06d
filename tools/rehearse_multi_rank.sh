set -o pipefail
export VRT_BENCH_BACKEND=gloo VRT_BENCH_VERIFY=1
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline "${@:3}" 2>gpurun_out/rehearse.err | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    o=json.loads(l); print(o['n_gpus'], o['scaling'], o['config']['width'], o['config']['height'], o['config']['parallelism'], o['config']['output'][:5], 'K', o['config']['streams'], 'G', o['config']['frames_per_launch'], 'equal:', o.get('assembled_frame_equals_single_gpu_frame'), 'other:', (o.get('other_exchange') or {}).get('last_frame_equals_single_gpu_frame'), 'c4:', (o.get('config4') or {}).get('assembled_frame_equals_single_gpu_frame'), 'Mrays/s', o['value'])
" || { echo FAILED "$@"; tail -5 gpurun_out/rehearse.err; }; }
run 2 29511 --workload c3
run 2 29518 --workload c3 --exchange gather
# buffers are reused (more blocks than streams), the last round is dealt unevenly
STEPS=5 run 2 29516 --workload c3 --frames-in-flight 3 --block-frames 2 --frames-per-step 29
STEPS=5 run 2 29517 --workload c2
run 3 29512 --workload c3 --frames-in-flight 3
run 2 29513 --workload c4 --scaling strong
run 2 29514 --workload c5 --strip-rows 0
run 4 29515 --workload c2 --output f32 --frames-in-flight 1
