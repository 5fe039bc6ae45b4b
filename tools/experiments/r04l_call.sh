# r04l: which part of r04k's fused BxDF-list evaluation costs the all-Lambert scene (config 3 under PG_FORCE_EXT=1: 76.9 -> 93.4 ms of
# k_shade<1> per frame) what it gains on microfacet scenes: each part switched off alone (fl0: f and pdf for the light's direction in two
# walks; fs0: Sample_f's pdf and f loops apart; sd0: the chosen BxDF's dead f computed), against the fused default and r04j's packed build.
OUT=gpurun_out/r04l; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # name lib env workload-args...
  n=$1; v=$2; e=$3; shift 3
  L=$PWD/pbrt-v3_amd/libpbrt_gpu.so; [ $v != default ] && L=$PWD/gpurun_in_libpbrt_gpu_$v.so
  ( env $e PBRT_GPU_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc --out $OUT/${n}_$v.pfm "$@" 2> $OUT/${n}_$v.err ) > $OUT/${n}_$v.json
  python - $n $v $OUT <<'PY'
import json,sys,hashlib
w,v,out=sys.argv[1:]
try:
    d=json.load(open(f"{out}/{w}_{v}.json"))
    print(w,v,round(d["value"],1),"Mrays/s",round(d["ms_per_step"],1),"ms",{k:round(x,1) for k,x in d["kernel_ms_per_step"].items()},"image",hashlib.md5(open(f"{out}/{w}_{v}.pfm","rb").read()).hexdigest()[:12])
except Exception as e: print(w,v,"FAILED",e)
PY
  rm -f $OUT/${n}_$v.pfm
}
for v in ${VARIANTS:-fl0 fs0 sd0}; do
  run cfg3ext $v PG_FORCE_EXT=1
  run div5m $v A=1 --workload divergent --tris 5000000 --spp 64
done
