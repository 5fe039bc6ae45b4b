#!/bin/bash
# What binds k_shade / k_trace?  Separate rocprofv3 --pmc passes (never combined with traces) over one reduced-spp frame:
# instruction-cache, scalar-cache, texture-addresser (TA), vector L1 (TCP) stall and translation counters, LDS conflicts,
# atomics at the L2.  usage: bash tools/pmc_diag.sh TAG [bench args]
TAG=${1:-diag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --no-cpu-baseline --spp 4}"
i=0
for C in \
  "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" \
  "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD" \
  "TCC_ATOMIC_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_BUSY_avr" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $C" >> $OUT/failed.txt
done
python - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("k_trace", "k_shade", "k_resolve", "k_generate")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k in sorted(tot):
        c = tot[k]
        fo.write(f"{k}  ({len(n[k])} dispatch ids)\n")
        for name in sorted(c): fo.write(f"    {name:40s} {c[name]:.5g}\n")
        def ratio(a, b, label):
            if c.get(a) is not None and c.get(b): fo.write(f"    -- {label}: {c[a] / c[b]:.4f}\n")
        ratio("SQC_ICACHE_MISSES", "SQC_ICACHE_REQ", "instruction-cache miss rate")
        ratio("SQC_DCACHE_MISSES", "SQC_DCACHE_REQ", "scalar-cache miss rate")
        ratio("SQ_IFETCH_LEVEL", "SQ_IFETCH", "mean instruction-fetch latency (counter units)")
        ratio("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum", "L1 TLB miss rate")
        ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "mean L1->L2 read latency (cycles)")
        ratio("TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP pending-stall share of TCP-active cycles")
        ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "LDS bank-conflict share")
print(open(os.path.join(out, "summary.txt")).read())
PY
cat $OUT/failed.txt 2>/dev/null
find $OUT -name '*.csv' -size +2M -delete
