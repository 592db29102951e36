set -x
O=gpurun_out/r3b; mkdir -p $O
cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o mfma4x4_bcast mfma4x4_bcast.hip && ./mfma4x4_bcast > ../../$O/micro.txt 2>&1; cd ../..
head -3 $O/micro.txt
timeout 2000 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
python tools/sbench.py --batch 1 > $O/sbench_b1.txt 2>&1
python tools/sbench.py --batch 8 > $O/sbench_b8.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 > $O/sbench_b8_kitti.txt 2>&1
cat $O/sbench_*.txt
python tools/pool_bench.py > $O/pool.txt 2>&1; cat $O/pool.txt
GPU_MAX_HW_QUEUES=8 python tools/pool_bench.py --workers 3,4,6 > $O/pool_q8.txt 2>&1; cat $O/pool_q8.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_s20.json 2> $O/bench_s20.err
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined > $O/bench_s200.json 2> $O/bench_s200.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tr1.json 2> $O/bench_tr1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 40 --warmup 5 --batch 8 --no-cpu-baseline > $O/bench_tr1_b8.json 2> $O/bench_tr1_b8.err
for i in 1 2 4; do python tools/rbench.py --batch $i --iters 30 2>&1 | grep -A4 "ref_order=0" > $O/rbench_b$i.txt; done; cat $O/rbench_b*.txt
python tools/stamps.py mid8q3 8 > $O/stamps_mid8q3_b8.txt 2>&1; python tools/stamps.py mid8q3 1 > $O/stamps_mid8q3_b1.txt 2>&1; cat $O/stamps_mid8q3_b*.txt
python -m lwsnet_amd.build --force > /dev/null 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('collective'))
    except Exception as e: print(f, 'ERR', e)
"
