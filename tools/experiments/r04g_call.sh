OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc 2>/dev/null ) > $OUT/$name.json; python - $name $OUT <<'PY'
import json,sys
n,o=sys.argv[1:]
d=json.load(open(f"{o}/{n}.json"))
print(n, round(d["value"],1), "Mrays/s", {k["kernel"].split(" ")[0]:round(k["avg_launch_ms"],2) for k in d["roofline_kernels"]})
PY
}
run base A=1
run depth16_5waves PG_TRACE_DEPTH=16
run depth8_5waves PG_TRACE_DEPTH=8 PG_TRACE_LDS_PAD=16384
run depth5_5waves PG_TRACE_DEPTH=5 PG_TRACE_LDS_PAD=22528
run depth8_8waves PG_TRACE_DEPTH=8
run depth5_8waves PG_TRACE_DEPTH=5
