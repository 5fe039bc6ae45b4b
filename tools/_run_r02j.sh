OUT=gpurun_out/r02j; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$OUT/$name.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})"
}
run base X=1
run b128 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_b128.so
run b64 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_b64.so
