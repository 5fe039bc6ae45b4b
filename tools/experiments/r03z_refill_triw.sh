#!/bin/bash
# r03z: k_trace<0, 0> issues 0.45 of the chip's vector-instruction slots (profiles/r03y_pmc_cfg3_valu.txt) at 55 % lane use: the two knobs that
# trade lane use against extra iterations -- the refill threshold (idle lanes before a wave refills) and the interior : leaf weighting.
OUT=gpurun_out/r03z; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> $OUT/$name.err ) > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1]))
    ks = {k["kernel"].split(" ")[0]: round(k["avg_launch_ms"], 2) for k in b["roofline_kernels"]}
    print(sys.argv[2], round(b["value"], 1), "Mrays/s", round(b["ms_per_step"], 1), "ms", ks)
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run default X=1
for r in 4 8 12 24 32; do run refill$r PG_TRACE_REFILL=$r; done
for w in 4 6 12 16 24; do run triw$w PG_TRACE_TRIW=$w; done
run refill8_triw12 PG_TRACE_REFILL=8 PG_TRACE_TRIW=12
run seg64 PG_TRACE_SEG=64
run seg256 PG_TRACE_SEG=256
