#!/bin/bash
# Round 3, GPU call 8: the whole quick GPU suite on the build with subsurface scattering (path + volpath) and the free-order shadow rays,
# the headline line, the divergent stand-ins.
OUT=gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 2 2> $OUT/bench.err ) > $OUT/bench.json; cut -c1-700 $OUT/bench.json
python - <<PY
import json
j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('kernel ms/step', {a:round(b,1) for a,b in j['kernel_ms_per_step'].items()})
print('hbm_regime', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['roofline'].get('hbm_regime',{}).items() if k in ('achieved','frac','avg_launch_ms','Mrays_per_s','ms_per_step')})
print('cpu', j.get('cpu_baseline'))
PY
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
run div5m timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64
run div10m_vol timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent-vol --tris 10000000 --spp 32
