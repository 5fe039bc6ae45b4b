#!/bin/bash
# r06n: the shading kernels' register peak (the BSDF sample of EstimateDirect's MIS half) relieved by hand: the vertex's pending light terms stored before it,
# beta parked in LDS across it, the ray's direction read again after it -- k_shade<1 | 3, VOL> 168 + 11 / 13 spilled -> 167, no scratch.  old = the library of r06l
OUT=gpurun_out/${1:-r06n}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_material_prepass.py tests/test_gpu_shade_order.py tests/test_gpu_parity.py tests/test_grid_medium.py tests/test_subsurface.py -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
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
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C0="timeout 300 python bench.py --workload config0 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
SV="timeout 400 python bench.py --steps 2 --warmup 1 --workload synthetic-vol --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
for v in new old new old; do
L=""; [ $v = old ] && L="PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prev.so"
run c3_$v $L $C3
run div5m_$v $L $DIV
run div10mvol_$v $L $VOL
run synvol_$v $L $SV
run config0_$v $L $C0
done
} | tee $OUT/ab.txt
