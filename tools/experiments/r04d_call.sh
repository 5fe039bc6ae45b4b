OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
for v in default s2w3 s2w4; do
  L=$PWD/pbrt-v3_amd/libpbrt_gpu.so; [ $v != default ] && L=$PWD/gpurun_in_libpbrt_gpu_$v.so
  ( PBRT_GPU_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline 2> /dev/null ) > $OUT/div5m_$v.json
  ( PBRT_GPU_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline 2> /dev/null ) > $OUT/div10m_$v.json
  python - $v $OUT <<'PY'
import json,sys
v,out=sys.argv[1:]
for w in ("div5m","div10m"):
    d=json.load(open(f"{out}/{w}_{v}.json"))
    ks={k["kernel"].split(" ")[0]:round(k["avg_launch_ms"],2) for k in d["roofline_kernels"]}
    print(v,w,round(d["value"],1),"Mrays/s",round(d["ms_per_step"],1),"ms",ks)
PY
done
for seg in 128 64 32; do PG_TRACE_SEG=$seg timeout 300 python tools/shard_timing.py > $OUT/shard_timing_seg$seg.json 2>/dev/null; python -c "
import json;d=json.load(open('$OUT/shard_timing_seg$seg.json'));print('seg $seg',{n:(v['render_ms'],v['speedup_bound']) for n,v in d['shards'].items()})"; done
