#!/bin/bash
# r06l: per-vertex normals + (u, v) in ONE 64-byte record per triangle (DScene::triAttr) against two arrays (48 B + 24 B per triangle, straddling lines)
OUT=gpurun_out/${1:-r06l}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_material_prepass.py tests/test_gpu_shade_order.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:18s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
    for kn,v in (j.get('pmc_by_kernel') or {}).items():
        if any(t in kn for t in ('k_shade','k_material')): print(f"    {kn:36s} fetch {v['fetch_KiB_per_launch']*2*1024/1e9:6.2f} GB  write {v['write_KiB_per_launch']*1024/1e9:6.2f} GB per launch  L2 hit {v['l2_hit_rate']:.3f}")
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
DIVP="timeout 900 python bench.py --steps 1 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-hbm-regime"
{
for v in new old new old; do
L=""; [ $v = old ] && L="PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prev.so"
run div5m_$v $L $DIV
run div10mvol_$v $L $VOL
run c3_$v $L $C3
done
run div5m_pmc_new $DIVP
run div5m_pmc_old PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prev.so $DIVP
} | tee $OUT/ab.txt
