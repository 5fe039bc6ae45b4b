OUT=gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/$name.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})"
}
run base X=1
run aux PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_aux.so
run base_refill8 PG_TRACE_REFILL=8
run aux_refill8 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_aux.so PG_TRACE_REFILL=8
run base2 X=1
run aux2 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_aux.so
