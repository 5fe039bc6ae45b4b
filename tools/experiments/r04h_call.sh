# r04h: materials evaluated ahead of the shading launch (k_material + k_shade<3>) against the evaluation inside the kernel (k_shade<2>, the
# library of the commit before), and k_material compiled for 3 / 4 / 5 / 6 waves per SIMD.  One gpurun call; images compared bit for bit.
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # name lib workload tris spp
  L=$PWD/pbrt-v3_amd/libpbrt_gpu.so; [ $2 != default ] && L=$PWD/gpurun_in_libpbrt_gpu_$2.so
  ( PBRT_GPU_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --workload $3 --tris $4 --spp $5 --no-cpu-baseline --no-hbm-regime --no-live-pmc --out $OUT/$1_$2.pfm 2> $OUT/$1_$2.err ) > $OUT/$1_$2.json
  python - $1 $2 $OUT <<'PY'
import json,sys,hashlib
w,v,out=sys.argv[1:]
try:
    d=json.load(open(f"{out}/{w}_{v}.json"))
    ks={k["kernel"].split(" ")[0]:round(k["avg_launch_ms"],2) for k in d["roofline_kernels"]}
    print(w,v,round(d["value"],1),"Mrays/s",round(d["ms_per_step"],1),"ms",ks,"image",hashlib.md5(open(f"{out}/{w}_{v}.pfm","rb").read()).hexdigest()[:12])
except Exception as e: print(w,v,"FAILED",e)
PY
  rm -f $OUT/$1_$2.pfm
}
for v in base default mw3 mw5 mw6; do run div5m $v divergent 5000000 64; done
for v in base default; do run div10m $v divergent-vol 10000000 32; done
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_div5m -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --no-live-pmc --workload divergent --tris 5000000 --spp 64 > $OUT/bench_prof_div5m.json 2> $OUT/prof_div5m.err ); find $OUT/prof_div5m -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_div5m.csv \; ; rm -rf $OUT/prof_div5m; head -12 $OUT/kernel_stats_div5m.csv | cut -c1-200
