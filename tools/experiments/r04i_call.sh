# r04i: the BxDF-list shading body (k_shade<1> / <3>) under other occupancy / instruction-scheduling choices, on the divergent stand-in
# and on config 3 with PG_FORCE_EXT=1 (k_shade<1> over one Lambert lobe).  One gpurun call.
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # name lib env workload-args...
  n=$1; v=$2; e=$3; shift 3
  L=$PWD/pbrt-v3_amd/libpbrt_gpu.so; [ $v != default ] && L=$PWD/gpurun_in_libpbrt_gpu_$v.so
  ( env $e PBRT_GPU_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc "$@" 2> $OUT/${n}_$v.err ) > $OUT/${n}_$v.json
  python - $n $v $OUT <<'PY'
import json,sys
w,v,out=sys.argv[1:]
try:
    d=json.load(open(f"{out}/{w}_{v}.json"))
    print(w,v,round(d["value"],1),"Mrays/s",round(d["ms_per_step"],1),"ms",{k["kernel"].split(" ")[0]:round(k["avg_launch_ms"],2) for k in d["roofline_kernels"]})
except Exception as e: print(w,v,"FAILED",e)
PY
}
for v in ${VARIANTS:-mw3 sw4 ilp mmc}; do
  run div5m $v A=1 --workload divergent --tris 5000000 --spp 64
  run cfg3ext $v PG_FORCE_EXT=1
done
