OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})"
}
VOL="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32"
EXT="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --spp 32"
run vol_w0 $VOL
run vol_w3 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w3.so $VOL
run vol_w4 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w4.so $VOL
run vol_w0b $VOL
run ext_w0 PG_FORCE_EXT=1 $EXT
run ext_w3 PG_FORCE_EXT=1 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w3.so $EXT
run ext_w4 PG_FORCE_EXT=1 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w4.so $EXT
