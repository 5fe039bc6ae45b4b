#!/bin/bash
# Round 3, GPU call 5: (1) k_shade<2, .> compiled for 2 / 3 waves per SIMD (1 today: 255 + 3 registers, two over the edge);
# (2) VERDICT r02 item 4 (iii): any-hit traversal in free order (nearer child first, no visit bookkeeping) -- films must equal the
# exact-order library's bit for bit; what does it buy?
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/gpurun_in_libpbrt_gpu
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
D5="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64"
D10="timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent-vol --tris 10000000 --spp 32"
C3="timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime"
run div5m_s2w1 $D5
run div5m_s2w2 PBRT_GPU_LIB=${L}_s2w2.so $D5
run div5m_s2w3 PBRT_GPU_LIB=${L}_s2w3.so $D5
run div10m_vol_s2w1 $D10
run div10m_vol_s2w2 PBRT_GPU_LIB=${L}_s2w2.so $D10
run div10m_vol_s2w3 PBRT_GPU_LIB=${L}_s2w3.so $D10
( timeout 600 python tools/experiments/r03e_films.py $OUT/films_exact.npz 2>&1 | tail -1 )
( PBRT_GPU_LIB=${L}_anyfree.so timeout 600 python tools/experiments/r03e_films.py $OUT/films_free.npz 2>&1 | tail -1 )
python - <<PY
import numpy as np
a, b = np.load('$OUT/films_exact.npz'), np.load('$OUT/films_free.npz')
bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
print('any-hit free order: films of', len(a.files) // 2, 'scenes', 'ALL bit-identical to the exact order' if not bad else ('DIFFER: ' + ' '.join(bad[:10])))
PY
rm -f $OUT/films_exact.npz $OUT/films_free.npz
run cfg3_exact $C3
run cfg3_anyfree PBRT_GPU_LIB=${L}_anyfree.so $C3
run div5m_anyfree PBRT_GPU_LIB=${L}_anyfree.so $D5
run cfg3_exact_b $C3
