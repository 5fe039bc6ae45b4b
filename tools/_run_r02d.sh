OUT=gpurun_out/r02d; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "sharded" 2>&1 | tail -15 ) > $OUT/pytest_sharded.log
tail -5 $OUT/pytest_sharded.log
# N = 2 over RCCL with both ranks on the one GPU of this box (functional pre-flight only)
( PBRT_BENCH_OVERSUBSCRIBE=1 NCCL_DEBUG=WARN timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --spp 8 --out $OUT/n2.pfm > $OUT/bench_n2.json 2> $OUT/bench_n2.err ); echo "n2 rc=$?"
tail -3 $OUT/bench_n2.json | cut -c1-600; tail -8 $OUT/bench_n2.err | cut -c1-300
( timeout 240 python bench.py --gpus 1 --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --out $OUT/n1.pfm > $OUT/bench_n1.json 2> $OUT/bench_n1.err ); python -c "
import numpy as np,sys
sys.path.insert(0,'.')
from __graft_entry__ import load_package
p=load_package()
try:
    a=p.read_pfm('$OUT/n1.pfm'); b=p.read_pfm('$OUT/n2.pfm'); print('n1 vs n2 identical:', np.array_equal(a,b))
except Exception as e: print('compare failed', e)
"
