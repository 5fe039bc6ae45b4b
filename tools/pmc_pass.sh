#!/bin/bash
# PMC passes over a short bench run (separate passes: counters never combined with --stats / traces).
# usage: bash tools/pmc_pass.sh TAG [bench args...]
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${@:---steps 1 --warmup 0 --spp 8 --no-cpu-baseline --no-hbm-regime}"
[ -f gpurun_out/counters.txt ] || rocprofv3 -L > gpurun_out/counters.txt 2>&1
i=0
MAXP=${PMC_PASSES:-99}
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  [ $i -gt $MAXP ] && break
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -o pmc -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
done
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +4M -delete
