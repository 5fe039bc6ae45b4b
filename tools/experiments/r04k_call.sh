# r04k: BSDF::f and BSDF::Pdf of a BxDF list in one walk (lobe_f_pdf: the microfacet half vector, D and Lambda(wo) once for both),
# non-specular Sample_f values not computed where BSDF::Sample_f discards them -- against r04j's packed build.
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
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
for v in ${VARIANTS:-packed default}; do
  run div5m $v A=1 --workload divergent --tris 5000000 --spp 64
  run div10m $v A=1 --workload divergent-vol --tris 10000000 --spp 32
  run cfg3ext $v PG_FORCE_EXT=1
done
for v in ${VARIANTS:-packed default}; do run config0 $v A=1 --workload config0 --spp 64; done
