#!/bin/bash
# r06o: what the integrator statistics of ABI 28 cost (per-wave ballots / reductions + <= 4 atomics at the end of k_shade / k_sss_exit, two ballots in k_resolve).
# First call: the first version (four atomics + three wave reductions per wave) against r06n's table; second call ("ab"): new = one packed 64-bit add per wave,
# shortest / longest only from lanes that improve on their shard, old = the first version (gpurun_in_libpbrt_gpu_prev.so).
OUT=gpurun_out/${1:-r06o}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
[ "$SKIP_TESTS" = 1 ] || ( PBRT_SKIP_SLOW=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
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
