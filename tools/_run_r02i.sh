OUT=gpurun_out/r02i; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
( timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$OUT/bench.err ) > $OUT/bench.json; python -c "
import json
j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in j['kernel_ms_per_step'].items()})"
