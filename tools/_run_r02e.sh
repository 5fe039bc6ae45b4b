OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sparse or many_lights" 2>&1 | tail -25 ) > $OUT/pytest_sparse.log
tail -12 $OUT/pytest_sparse.log
( PBRT_BENCH_OVERSUBSCRIBE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --spp 8 --out $OUT/n2.pfm > $OUT/bench_n2.json 2> $OUT/bench_n2.err ); echo "n2 rc=$?"
tail -1 $OUT/bench_n2.json | cut -c1-400; tail -4 $OUT/bench_n2.err | cut -c1-300
( timeout 240 python bench.py --gpus 1 --steps 1 --warmup 1 --spp 8 --no-cpu-baseline --out $OUT/n1.pfm > $OUT/bench_n1.json 2> $OUT/bench_n1.err ); python -c "
import numpy as np,sys
sys.path.insert(0,'.')
from __graft_entry__ import load_package
p=load_package()
try:
    a=p.read_pfm('$OUT/n1.pfm'); b=p.read_pfm('$OUT/n2.pfm'); print('n1 vs n2 identical:', np.array_equal(a,b))
except Exception as e: print('compare failed', e)
"
rm -f $OUT/n1.pfm $OUT/n2.pfm
