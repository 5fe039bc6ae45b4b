# r04m: instruction-scheduling strategies of the compiler for the shading translation unit (pg_kernels.hip: max-ilp, max-memory-clause,
# early if-conversion) and for the traversal one (pg_traverse.hip: max-memory-clause; max-ilp takes k_trace<0,0> to 83 VGPRs = 5 waves,
# a known loss, and was not run), on config 3 and the 5 M divergent stand-in against the default build.
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
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
for v in default ilp mmc eif mmctr; do run cfg3 $v A=1; done
for v in default ilp mmc eif; do run div5m $v A=1 --workload divergent --tris 5000000 --spp 64; done
