#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE for k_trace's access pattern: pbrt-v3_amd/ubench_gather --calib runs
# one launch of the random 64-B record gather and one coalesced 16 B/lane streaming read over a table far larger than the
# Infinity Cache, both with exactly known byte counts; FETCH_SIZE and the L2 hit / miss counters are taken in separate --pmc
# passes.  Prints, per kernel, requested bytes, FETCH_SIZE bytes, the L2 hit rate and factor = requested x (1 - hit) / FETCH_SIZE
# -- what FETCH_SIZE has to be multiplied by for this pattern (the guide's x2 is for the streaming read).
# usage (GPU box): bash tools/pmc_calibrate.sh TAG [MiB]
TAG=${1:-calib}; MB=${2:-925}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$D -o pmc -- pbrt-v3_amd/ubench_gather --calib $MB > $OUT/$D.json 2> $OUT/$D.err
done
python - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
known = json.loads(open(os.path.join(out, "FETCH_SIZE.json")).read().strip().splitlines()[-1])
val = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        val.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"table_MiB": known["table_MiB"], "kernels": {}}
for k, req in (("k_gather<0>", known["gather_requested_bytes"]), ("k_stream", known["stream_bytes"]), ("k_gather_pair", 2 * known["gather_requested_bytes"])):
    v = next((c for n, c in val.items() if n.replace("void ", "").startswith(k)), None)
    if not v: continue
    fetch = v.get("FETCH_SIZE", 0.0) * 1024
    h = v.get("TCC_HIT_sum", 0.0) / max(1.0, v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))
    res["kernels"][k] = {"requested_bytes": req, "FETCH_SIZE_bytes": fetch, "l2_hit_rate": h,
                         "factor_requested_missed_over_FETCH_SIZE": req * (1 - h) / fetch if fetch else None,
                         "factor_requested_over_FETCH_SIZE": req / fetch if fetch else None,
                         "TCC_HIT": v.get("TCC_HIT_sum"), "TCC_MISS": v.get("TCC_MISS_sum"),
                         "TCC_EA0_RDREQ": v.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": v.get("TCC_EA0_RDREQ_32B_sum")}
json.dump(res, open(os.path.join(out, "fetch_size_calibration.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name '*.csv' -size +4M -delete
