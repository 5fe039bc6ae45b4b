#!/bin/bash
# rocprofv3 kernel stats of one bench run per library variant: usage prof_variant.sh TAG lib1.so lib2.so ...
TAG=$1; shift
export TMPDIR=/tmp
for LIB in "$@"; do
  N=$(basename $LIB .so); OUT=gpurun_out/$TAG/$N; mkdir -p $OUT
  PBRT_GPU_LIB=$PWD/$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py ${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline} > $OUT/bench.json 2> $OUT/err
  echo "== $N: $(python -c "import json;j=json.load(open('$OUT/bench.json'));print(round(j['value'],1),'Mrays/s', round(j['ms_per_step'],1),'ms')")"
  cut -d, -f1-4 $OUT/t_kernel_stats.csv | sed 's/(.*)"/"/' | head -7
  find $OUT -name '*kernel_trace*' -delete
done
