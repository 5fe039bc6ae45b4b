#!/bin/bash
# r06f: guided self-scheduling of a region's last round in k_trace: what a 1/8 film shard costs on one GPU (tools/shard_timing.py), whole frames unchanged?
OUT=gpurun_out/${1:-r06f}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
for g in 0 16 32 64; do
  ( PG_TRACE_GUIDED=$g timeout 300 python tools/shard_timing.py > $OUT/shard_g$g.json 2> $OUT/shard_g$g.err )
  python - $g $OUT/shard_g$g.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2])); s=j["shards"]
print(f"guided {sys.argv[1]:>3}: " + "  ".join(f"1/{n}: {s[n]['render_ms']:.2f} ms (closest {s[n]['kernels_ms']['closest_ms']:.2f}, shadow {s[n]['kernels_ms']['shadow_ms']:.2f}) bound {s[n]['speedup_bound']}" for n in ("1","2","4","8")))
PY
done | tee $OUT/ab.txt
for g in 0 32; do
  ( SHARD_OVERLAP=1 PG_TRACE_GUIDED=$g timeout 300 python tools/shard_timing.py > $OUT/shard_overlap_g$g.json 2> $OUT/shard_overlap_g$g.err )
  python - $g $OUT/shard_overlap_g$g.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2])); s=j["shards"]
print(f"overlap, guided {sys.argv[1]:>3}: " + "  ".join(f"1/{n}: {s[n]['render_ms']:.2f} ms bound {s[n]['speedup_bound']}" for n in ("1","2","4","8")))
PY
done | tee -a $OUT/ab.txt
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "shard or tile or instance_rays or fullsize or golden_images and (cornell_32 or instance_boxes or divergent_small)" 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $OUT/pytest.log; cat $OUT/pytest.log
( timeout 600 python bench.py --steps 3 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_div5m.err ) > $OUT/bench_div5m.json; cut -c1-400 $OUT/bench_div5m.json
