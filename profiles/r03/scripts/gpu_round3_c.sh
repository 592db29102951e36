set -x
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "conv3d_stack or schedule_options or disparity_stages" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python tools/sbench.py --batch 1 > $O/sbench_b1.txt 2>&1
python tools/sbench.py --batch 8 > $O/sbench_b8.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 > $O/sbench_b8_kitti.txt 2>&1
cat $O/sbench_*.txt
cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o copybw copybw.hip && ./copybw 8 > ../../$O/copybw_b8.txt 2>&1; ./copybw 1 > ../../$O/copybw_b1.txt 2>&1; cd ../..
cat $O/copybw_b8.txt $O/copybw_b1.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_q4_s20.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_q8_s20.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_q8_s200.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_form=1 > $O/bench_q8_s200_q.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt side_streams=0 > $O/bench_q8_s200_ss0.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tr1_q4.json 2> $O/bench_tr1.err
GPU_MAX_HW_QUEUES=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tr1_q8.json 2> $O/bench_tr1_q8.err
GPU_MAX_HW_QUEUES=8 python tools/pool_bench.py --workers 3,4,5 --opt mid8_form=1 > $O/pool_q8_mid8q.txt 2>&1; cat $O/pool_q8_mid8q.txt
GPU_MAX_HW_QUEUES=16 python tools/pool_bench.py --workers 4,6 > $O/pool_q16.txt 2>&1; cat $O/pool_q16.txt
python tools/stamps.py mid8q3 8 > $O/stamps_mid8q3_b8.txt 2>&1; python tools/stamps.py mid8q2 8 > $O/stamps_mid8q2_b8.txt 2>&1; python tools/stamps.py mid8_3 8 > $O/stamps_mid8_3_b8.txt 2>&1; cat $O/stamps_*.txt
python -m lwsnet_amd.build --force > /dev/null 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], (d.get('pipelined') or {}).get('value'), d.get('collective'))
    except Exception as e: print(f, 'ERR', e)
"
