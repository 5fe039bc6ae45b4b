#!/bin/bash
# HBM-side bytes of every kernel from PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (no --stats/trace), units KiB, read side doubled (gfx950: FETCH_SIZE tallies 128-B requests at 64 B;
# tools/pmc_calibrate.sh shows the same for k_trace's 64-B record gather: an L2 miss fills the whole 128-B line).
# A third pass takes the L2 hit/miss counters, a fourth the vector-instruction counters (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU).  Merges the result into OUT/pmc_traffic.json under the workload's name; copy
# that file to profiles/pmc_traffic.json and bench.py reports it as roofline.traffic for the same workload.
# usage: bash tools/pmc_traffic.sh TAG [bench args]
TAG=${1:-traffic}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime}"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
  D=$(echo $C | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/$D -o pmc -- python bench.py $ARGS > $OUT/$D.json 2> $OUT/$D.err
done
python - $OUT "$ARGS" <<'PY'
import csv, glob, json, os, sys
out, args = sys.argv[1], sys.argv[2]
tot = {}
for d, names in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum", ["TCC_HIT_sum", "TCC_MISS_sum"]),
                 ("SQ_INSTS_VALU", ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"])):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c not in names: continue
            k = r["Kernel_Name"].split("(")[0]
            e = tot.setdefault(c, {}).setdefault(k, [0.0, set()])
            e[0] += float(r["Counter_Value"]); e[1].add(r["Dispatch_Id"])
j = json.loads(open(os.path.join(out, "FETCH_SIZE.json")).read().strip().splitlines()[-1])
res = {"bench_args": args, "kernels": {}}
for k, (f, nf) in tot.get("FETCH_SIZE", {}).items():
    w, nw = tot.get("WRITE_SIZE", {}).get(k, (0.0, {0}))
    e = {"launches": len(nf), "fetch_KiB_per_launch": f / len(nf), "write_KiB_per_launch": w / max(1, len(nw)),
         "hbm_bytes_per_launch": (2 * f / len(nf) + w / max(1, len(nw))) * 1024}
    h, m = tot.get("TCC_HIT_sum", {}).get(k, (0.0, 0))[0], tot.get("TCC_MISS_sum", {}).get(k, (0.0, 0))[0]
    if h + m > 0: e["l2_hit_rate"] = h / (h + m)
    vi, vt = tot.get("SQ_INSTS_VALU", {}).get(k, (0.0, {0})), tot.get("SQ_THREAD_CYCLES_VALU", {}).get(k, (0.0, 0))[0]
    if vi[0] > 0:  # vector (wave-wide) instructions per launch and the lanes active per instruction: what the SIMDs issued
        e["valu_insts_per_launch"] = vi[0] / max(1, len(vi[1])); e["valu_lanes_active"] = vt / vi[0]
    res["kernels"][k] = e
res["note"] = ("FETCH_SIZE, WRITE_SIZE in KiB from separate rocprofv3 --pmc passes; hbm_bytes_per_launch = 2 x read + write (a counted read request moves "
               "a 128-B line: calibrated for streams and for the record gather, profiles/fetch_size_calibration.json); the L2's memory-side requests, "
               "Infinity-Cache hits included")
dst = os.path.join(out, "pmc_traffic.json")
allj = {"workloads": {}}
for src in (os.path.join("profiles", "pmc_traffic.json"), dst):
    if os.path.exists(src):
        try: allj["workloads"].update(json.load(open(src)).get("workloads", {}))
        except Exception: pass
allj["workloads"][j["config"]["workload"]] = res
allj["round"] = os.path.basename(os.path.dirname(out.rstrip("/"))) or out  # gpurun_out/<round>/traffic_* -> <round>: bench.py quotes it with every replayed value
json.dump(allj, open(dst, "w"), indent=1)
print(json.dumps({j["config"]["workload"]: res}, indent=1))
PY
