#!/bin/bash
# Same-box A/B of the packed lobe records: the previous commit's device library (tools/experiments/_ab/libpbrt_gpu_prev.so, built from bf5a47e)
# against the tree's, alternating, on the two divergent stand-ins.  (The previous library is not kept in the tree: rebuild it from that commit to repeat.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export TMPDIR=/tmp
OUT=gpurun_out/r05o; mkdir -p $OUT
run() { # run TAG LIB bench-args
  local tag=$1 lib=$2; shift 2
  ( PBRT_GPU_LIB=$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc "$@" 2> $OUT/$tag.err ) > $OUT/$tag.json
  python - $OUT/$tag.json $tag <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
n=[r for r in d['roofline_kernels'] if r['kernel'].startswith('k_shade')][0]
print(sys.argv[2], round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms; shading slot', round(n['avg_launch_ms'],3), 'ms/launch x', n['launches'])
P
}
NEW=$PWD/pbrt-v3_amd/libpbrt_gpu.so; OLD=$PWD/tools/experiments/_ab/libpbrt_gpu_prev.so
for i in 1 2; do
  run div5m_prev_$i $OLD --workload divergent --tris 5000000 --spp 64
  run div5m_packed_$i $NEW --workload divergent --tris 5000000 --spp 64
  run vol_prev_$i $OLD --workload divergent-vol --tris 10000000 --spp 32
  run vol_packed_$i $NEW --workload divergent-vol --tris 10000000 --spp 32
done 2>&1 | tee $OUT/ab.txt
