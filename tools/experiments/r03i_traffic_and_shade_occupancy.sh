#!/bin/bash
# Round 3, GPU call: PMC traffic passes of the final kernels (config 3 and the 5 M-triangle HBM-regime workload at the headline's spp),
# then k_shade<1, .> compiled for 2 / 3 / 4 waves per SIMD (round 2's occupancy A/Bs were taken while one statistics word serialised the kernel).
OUT=gpurun_out/r03i; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_traffic.sh r03i/traffic_cfg3 > $OUT/traffic_cfg3.log 2>&1; tail -3 $OUT/traffic_cfg3.log | cut -c1-300
cp $OUT/traffic_cfg3/pmc_traffic.json profiles/pmc_traffic.json
bash tools/pmc_traffic.sh r03i/traffic_5m --steps 1 --warmup 0 --no-cpu-baseline --grid 1582 --spp 64 > $OUT/traffic_5m.log 2>&1; tail -3 $OUT/traffic_5m.log | cut -c1-300
cp $OUT/traffic_5m/pmc_traffic.json $OUT/pmc_traffic.json
find $OUT -name '*counter_collection.csv' -size +2M -delete
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
L=$PWD/gpurun_in_libpbrt_gpu
V="timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --grid 2237 --spp 32"
E="timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --spp 32"
run vol10m_w3 $V
run vol10m_w2 PBRT_GPU_LIB=${L}_s1w2.so $V
run vol10m_w4 PBRT_GPU_LIB=${L}_s1w4.so $V
run ext_w3 PG_FORCE_EXT=1 $E
run ext_w2 PG_FORCE_EXT=1 PBRT_GPU_LIB=${L}_s1w2.so $E
run ext_w4 PG_FORCE_EXT=1 PBRT_GPU_LIB=${L}_s1w4.so $E
