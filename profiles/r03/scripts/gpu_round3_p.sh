set -x
mkdir -p gpurun_out/r3p
python tools/split_bf16_numerics.py --pairs 8 > gpurun_out/r3p/numerics_64x256.txt 2>&1; tail -12 gpurun_out/r3p/numerics_64x256.txt
python tools/split_bf16_numerics.py --pairs 3 --size 256x512 > gpurun_out/r3p/numerics_256x512.txt 2>&1; tail -6 gpurun_out/r3p/numerics_256x512.txt
