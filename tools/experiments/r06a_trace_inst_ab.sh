#!/bin/bash
# r06a: step histogram of the current k_trace on the instanced stand-ins, then the A/B of the round-6 candidates for k_trace<., XP_INST | XP_ALPHA>
# (variants built by tools/build_variant.sh into gpurun_in_libpbrt_gpu_<name>.so).  One gpurun call.
OUT=gpurun_out/${1:-r06a}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_stats.so timeout 300 python tools/trace_step_stats.py --workload divergent --tris 5000000 --spp 16 > $OUT/stats_div5m.json 2> $OUT/stats_div5m.err
PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_stats.so timeout 300 python tools/trace_step_stats.py --workload divergent-vol --tris 10000000 --spp 8 > $OUT/stats_div10m_vol.json 2> $OUT/stats_div10m_vol.err
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:10s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
run base $DIV
for v in fastmod agrp hs all park nosz all2 w6 hs6 all6; do run $v PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$v.so $DIV; done
run all26_d10 PG_TRACE_DEPTH=10 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all26.so $DIV
run all2_d10 PG_TRACE_DEPTH=10 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all2.so $DIV
run base_b $DIV
run vol_base $VOL
for v in all2 all26; do run vol_$v PG_TRACE_DEPTH=10 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$v.so $VOL; done
} | tee $OUT/ab.txt
# parity of the combined variant on the scenes that reach the changed code
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all2.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_anyhit_order.py -m gpu -x -q -k "alpha or divergent or instance or motion or watertight or reintersect" 2>&1 | tail -5 ) > $OUT/pytest_all2.log
tail -3 $OUT/pytest_all2.log
