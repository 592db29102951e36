cd tools/micro && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o split_bf16 split_bf16.hip && ./split_bf16 | tee ../../gpurun_out/micro_split_bf16.txt
