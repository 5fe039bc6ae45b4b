#!/bin/bash
# r06g: VERDICT r05 item 4 -- the region a vertex's rays are appended to chosen by the hit point's cell (4 x 2 over the world bound's two longest axes)
# instead of by the producing block's XCD (PG_SPATIAL_REGIONS=1; no sort), on config 3, the 5 M heightfield and the two divergent stand-ins.
OUT=gpurun_out/${1:-r06g}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:18s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
M5="timeout 400 python bench.py --steps 2 --warmup 1 --grid 1582 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
for s in 0 1; do
run c3_spatial$s PG_SPATIAL_REGIONS=$s $C3
run m5_spatial$s PG_SPATIAL_REGIONS=$s $M5
run div5m_spatial$s PG_SPATIAL_REGIONS=$s $DIV
run div10mvol_spatial$s PG_SPATIAL_REGIONS=$s $VOL
done
} | tee $OUT/ab.txt
( PG_SPATIAL_REGIONS=1 PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $OUT/pytest_spatial.log; cat $OUT/pytest_spatial.log
