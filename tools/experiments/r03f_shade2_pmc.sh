#!/bin/bash
# Round 3, GPU call 6: what binds k_shade<2, false> (42 ms per launch on the divergent stand-in, no answer to occupancy)?  SQ / TCC
# counters per kernel (separate --pmc passes, no traces); the any-hit order tests on the device; free-order any-hit at 6 vs 5 waves.
OUT=gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_anyhit_order.py -m gpu -x -q 2>&1 | tail -3 ) > $OUT/pytest_anyhit_order.log; tail -2 $OUT/pytest_anyhit_order.log
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
D5="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64"
D10="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent-vol --tris 10000000 --spp 32"
run div5m $D5
run div5m_free5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_free5.so $D5
run div10m_vol $D10
bash tools/pmc_pass.sh r03f/pmc_div5m --steps 1 --warmup 0 --no-cpu-baseline --workload divergent --tris 5000000 --spp 16 > $OUT/pmc_div5m.log 2>&1
grep -A40 "k_shade" gpurun_out/r03f/pmc_div5m/summary.txt | head -60
