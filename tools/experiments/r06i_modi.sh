#!/bin/bash
# r06i: Mod() of every texel lookup (mip_texel: k_material, k_shade<2>, the alpha masks) skips its integer division where the coordinate is inside the map
OUT=gpurun_out/${1:-r06i}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:18s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
for v in new old new old; do
L=""; [ $v = old ] && L="PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prev.so"
run div5m_$v $L $DIV
run div10mvol_$v $L $VOL
done
} | tee $OUT/ab.txt
