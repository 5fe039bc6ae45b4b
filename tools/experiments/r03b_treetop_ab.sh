#!/bin/bash
# Round 3, GPU call 2: the A/B of call 1 again with the record loads behind address-space-typed pointers (call 1's build loaded
# through flat pointers: base 456 ms / frame in k_trace<false> instead of 219), ubench predictions, sibling-line calibration.
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime"
L=$PWD/gpurun_in_libpbrt_gpu
run base_k0_d11 $B
run b256_k127_d7 PG_TRACE_TOPK=127 PG_TRACE_DEPTH=7 $B
run b256_k63_d9 PG_TRACE_TOPK=63 PG_TRACE_DEPTH=9 $B
run b256_k31_d10 PG_TRACE_TOPK=31 PG_TRACE_DEPTH=10 $B
run b256_k255_d8 PG_TRACE_TOPK=255 PG_TRACE_DEPTH=8 $B
run b512_k0_d11 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=0 PG_TRACE_DEPTH=11 $B
run b512_k255_d9 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=255 PG_TRACE_DEPTH=9 $B
run b512_k127_d11 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=127 PG_TRACE_DEPTH=11 $B
run b512_k511_d5 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=5 $B
run b768_k511_d8 PBRT_GPU_LIB=${L}_b768.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=8 $B
run b768_k255_d10 PBRT_GPU_LIB=${L}_b768.so PG_TRACE_TOPK=255 PG_TRACE_DEPTH=10 $B
run b1024_k1023_d11 PBRT_GPU_LIB=${L}_b1024.so PG_TRACE_TOPK=1023 PG_TRACE_DEPTH=11 $B
run base_again $B
run stats_k127 PBRT_GPU_LIB=${L}_stats.so PG_TRACE_TOPK=127 PG_TRACE_DEPTH=7 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime; grep "k_trace<false> lanes" $OUT/stats_k127.err | tail -1
( timeout 300 pbrt-v3_amd/ubench_gather --predict 2 106 2>&1 ) > $OUT/ubench_predict.txt; cat $OUT/ubench_predict.txt
bash tools/pmc_calibrate.sh r03b/calib 925 > $OUT/calib.log 2>&1; grep -A12 "k_gather_pair" $OUT/calib.log | head -14
( PG_TRACE_TOPK=127 PG_TRACE_DEPTH=7 PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not golden_images and not sampler_" 2>&1 | tail -4 ) > $OUT/pytest_treetop.log; tail -2 $OUT/pytest_treetop.log
