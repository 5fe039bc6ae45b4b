#!/bin/bash
# Round 3, GPU call 1: (1) parity of the tree-top-in-LDS traversal on the real device, (2) the new bench line end to end (hbm_regime,
# cpu_baseline), (3) RCCL with one rank through the packed gather, (4) ubench predictions + FETCH_SIZE calibration,
# (5) A/B of tree-top size x stack depth x block size on config 3.
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
( PG_TRACE_TOPK=127 PG_TRACE_DEPTH=7 PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest_treetop.log
tail -3 $OUT/pytest_treetop.log
( timeout 600 python bench.py --steps 5 --warmup 2 2> $OUT/bench.err ) > $OUT/bench.json; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
( PBRT_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_rccl1.err ) > $OUT/bench_rccl1.json; cut -c1-700 $OUT/bench_rccl1.json; tail -2 $OUT/bench_rccl1.err
( timeout 300 pbrt-v3_amd/ubench_gather --predict 2 106 2>&1 ) > $OUT/ubench_predict.txt; cat $OUT/ubench_predict.txt
bash tools/pmc_calibrate.sh r03a/calib 925 > $OUT/calib.log 2>&1; tail -30 $OUT/calib.log
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
run b256_k255_d8 PG_TRACE_TOPK=255 PG_TRACE_DEPTH=8 $B
run b256_k31_d10 PG_TRACE_TOPK=31 PG_TRACE_DEPTH=10 $B
run b512_k0_d11 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=0 PG_TRACE_DEPTH=11 $B
run b512_k255_d9 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=255 PG_TRACE_DEPTH=9 $B
run b512_k511_d5 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=5 $B
run b512_k511_d8 PBRT_GPU_LIB=${L}_b512.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=8 $B
run b768_k511_d8 PBRT_GPU_LIB=${L}_b768.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=8 $B
run b1024_k1023_d11 PBRT_GPU_LIB=${L}_b1024.so PG_TRACE_TOPK=1023 PG_TRACE_DEPTH=11 $B
run b1024_k511_d6 PBRT_GPU_LIB=${L}_b1024.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=6 $B
run base_again $B
run stats_k127 PBRT_GPU_LIB=${L}_stats.so PG_TRACE_TOPK=127 PG_TRACE_DEPTH=7 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime; grep "k_trace<false> lanes" $OUT/stats_k127.err | tail -1
run stats_k511 PBRT_GPU_LIB=${L}_stats.so PG_TRACE_TOPK=511 PG_TRACE_DEPTH=4 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime; grep "k_trace<false> lanes" $OUT/stats_k511.err | tail -1
run stats_k2047 PBRT_GPU_LIB=${L}_stats.so PG_TRACE_TOPK=2047 PG_TRACE_DEPTH=2 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime; grep "k_trace<false> lanes" $OUT/stats_k2047.err | tail -1
