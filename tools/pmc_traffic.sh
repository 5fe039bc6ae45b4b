#!/bin/bash
# HBM-side bytes of the dominant kernel from PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (no --stats/trace), units KiB, read side doubled (gfx950: FETCH_SIZE tallies 128-B
# requests at 64 B).  Writes profiles/pmc_traffic.json, which bench.py reports as roofline.traffic for the same workload.
# usage: bash tools/pmc_traffic.sh TAG [bench args]
TAG=${1:-traffic}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --no-cpu-baseline}"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python bench.py $ARGS > $OUT/$C.json 2> $OUT/$C.err
done
python - $OUT "$ARGS" <<'PY'
import csv, glob, json, os, sys, re
out, args = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = {}; n = {}
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0]
            v[k] = v.get(k, 0.0) + float(r["Counter_Value"]); n.setdefault(k, set()).add(r["Dispatch_Id"])
    tot[c] = {k: (v[k], len(n[k])) for k in v}
j = json.loads(open(os.path.join(out, "FETCH_SIZE.json")).read().strip().splitlines()[-1])
res = {"workload": j["config"]["workload"], "kernels": {}}
for k in tot["FETCH_SIZE"]:
    f, nf = tot["FETCH_SIZE"][k]; w, nw = tot["WRITE_SIZE"].get(k, (0.0, 1))
    res["kernels"][k] = {"launches": nf, "fetch_KiB_per_launch": f / nf, "write_KiB_per_launch": w / max(1, nw),
                         "hbm_bytes_per_launch": (2 * f / nf + w / max(1, nw)) * 1024}
dom = [k for k in res["kernels"] if "k_trace<false" in k]
if dom: res["hbm_bytes_per_launch"] = res["kernels"][dom[0]]["hbm_bytes_per_launch"]
res["note"] = "FETCH_SIZE, WRITE_SIZE in KiB from separate rocprofv3 --pmc passes; read side x2 (gfx950 correction, calibrated for 16 B/lane streaming reads; an upper bound for this gather pattern)"
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
