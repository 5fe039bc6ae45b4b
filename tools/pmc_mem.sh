#!/bin/bash
# Memory-pipeline PMC passes (TA/TCP/TD busy and stall counters) over a short bench run.
TAG=${1:-pmcmem}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --spp 8 --no-cpu-baseline}"
i=0
for PMC in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TAGRAM0_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -o pmc -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
done
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete
