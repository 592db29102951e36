set -x
O=gpurun_out/r3f; mkdir -p $O
for rep in 1 2; do
python tools/sbench.py --batch 1 --opt conv3d_order=1 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 2 --opt conv3d_order=1 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --opt conv3d_order=1 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 --opt conv3d_order=1 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 1 --size 368x1232 --opt conv3d_order=1 >> $O/sbench.txt 2>&1
done
grep stage $O/sbench.txt
