set -x
O=gpurun_out/r3i; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
python tools/pool_bench.py --batch 4 --workers 2,3,4 --jobs 300 > $O/pool_b4.txt 2>&1; cat $O/pool_b4.txt
python tools/pool_bench.py --batch 8 --workers 2,3,4 --jobs 150 > $O/pool_b8.txt 2>&1; cat $O/pool_b8.txt
python tools/pool_bench.py --batch 2 --workers 2,4,6 --jobs 400 > $O/pool_b2.txt 2>&1; cat $O/pool_b2.txt
python tools/pool_bench.py --batch 1 --size 368x1232 --workers 2,4 --jobs 200 > $O/pool_kitti_b1.txt 2>&1; cat $O/pool_kitti_b1.txt
python tools/pool_bench.py --batch 4 --size 368x1232 --workers 2 --jobs 60 > $O/pool_kitti_b4.txt 2>&1; cat $O/pool_kitti_b4.txt
