#!/bin/bash
# Round 3, GPU call 4: the traversal kernels of instanced / alpha-masked scenes after the register diet (XP instantiations, world ray
# reloaded at instance exit, inline image-map alpha masks): parity on the device, then the divergent stand-ins again (r03c: 240.7 /
# 407.1 Mrays/s with the 196-VGPR general kernel at 2 waves per SIMD).
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not sampler_ and not 02sequence" 2>&1 | tail -4 ) > $OUT/pytest.log; tail -2 $OUT/pytest.log
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()}, j['config']['workload'][:100])
except Exception as e: print('$name FAILED', e)"
}
D5="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64"
D10="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent-vol --tris 10000000 --spp 32"
run cfg3 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime
run div5m_w6 $D5
run div5m_w5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w5.so $D5
run div5m_general PG_ALPHA_GENERAL=1 $D5
run div10m_vol_w6 $D10
run div10m_vol_w5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w5.so $D10
