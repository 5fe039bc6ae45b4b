#!/bin/bash
# Round 5, moving shapes / instances on the MI355X: the whole GPU suite (with the 11 motion_* goldens and 48 random moving scenes), 150 more random
# moving scenes, then the still workloads again (the shading kernels were recompiled: instance matrices through an accessor).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export TMPDIR=/tmp
OUT=gpurun_out/r05p; mkdir -p $OUT
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -30 ) > $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log
( timeout 300 python -u tools/fuzz_emulated_device.py 100 250 random_scene_motion 2>&1 | grep -v '^Warning' > $OUT/fuzz_motion.log ); tail -2 $OUT/fuzz_motion.log
run() { local tag=$1; shift
  ( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc "$@" 2> $OUT/$tag.err ) > $OUT/$tag.json
  python - $OUT/$tag.json $tag <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms;', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
P
}
run cfg3
run div5m --workload divergent --tris 5000000 --spp 64
run div10m_vol --workload divergent-vol --tris 10000000 --spp 32
run config0_64 --workload config0 --spp 64
