OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/$name.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})"
}
run head PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_head.so
run new X=1
run head2 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_head.so
run new2 X=1
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -4 )
