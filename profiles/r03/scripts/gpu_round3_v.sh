O=gpurun_out/r3v; mkdir -p $O
for o in "" "--opt side_streams=0" "--opt left_at=0" "--opt defer_upsample=0" "--opt fuse_first=0"; do
  for rep in 1 2; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined $o 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', d['value'], d['ms_per_step'])"
  done
done
