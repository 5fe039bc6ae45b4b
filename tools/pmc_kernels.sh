#!/bin/bash
# Per-kernel SQ counters of one reduced-spp frame (three --pmc passes, 8 SQ counters each): instruction mix, issue / wait
# cycles and the in-flight levels of the memory pipes.  usage: bash tools/pmc_kernels.sh TAG [bench args]
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --no-cpu-baseline --spp 8}"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"
i=0
for C in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
done
python - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not ("k_trace" in k or "k_shade" in k or "k_resolve" in k or "k_material" in k): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k in sorted(tot):
        c = tot[k]; w = max(1.0, c.get("SQ_WAVES", 1.0))
        fo.write(f"{k}  ({len(n[k])} dispatch ids)\n")
        for name in sorted(c): fo.write(f"    {name:28s} {c[name]:.4g}   per wave {c[name] / w:.4g}\n")
        if c.get("SQ_WAVE_CYCLES"):
            fo.write(f"    -- wait share of wave time {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}, issue-stall share {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}, "
                     f"active-lane share of VALU {c.get('SQ_THREAD_CYCLES_VALU', 0) / max(1.0, 64 * c.get('SQ_ACTIVE_INST_VALU', 1)):.3f}\n")
        if c.get("SQ_INSTS_VMEM_RD"): fo.write(f"    -- mean VMEM instructions in flight per wave-cycle: level/cycles = {c.get('SQ_INST_LEVEL_VMEM', 0) / max(1.0, c.get('SQ_WAVE_CYCLES', 1)):.3f}; latency ~ level / insts = {c.get('SQ_INST_LEVEL_VMEM', 0) / max(1.0, c.get('SQ_INSTS_VMEM_RD', 1) + c.get('SQ_INSTS_VMEM_WR', 0)):.1f} cycles\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
find $OUT -name '*counter_collection.csv' -size +2M -delete
