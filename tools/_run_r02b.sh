TAG=r02b; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
( timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/bench.err ) > $OUT/bench.json
( PG_HITTRI=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/bench_nohittri.err ) > $OUT/bench_nohittri.json
tail -6 $OUT/pytest_gpu.log
python - <<'PY'
import json
for f in ("bench","bench_nohittri"):
    try:
        j=json.loads(open(f"gpurun_out/r02b/{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["value"],1), "Mrays/s", round(j["ms_per_step"],1), "ms", {k:round(v,1) for k,v in j["kernel_ms_per_step"].items()})
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/r02b/{f}.err").read()[-500:])
PY
