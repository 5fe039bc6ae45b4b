# r05c: volpath shading with the path state in queue order -- one launch vs medium / surface parts (PG_VOL_PARTS); by slot: profiles r05b (parts0)
ulimit -c 0
mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
for p in 0 1; do
  ( PG_VOL_PARTS=$p timeout 300 python bench.py --workload divergent-vol --tris 10000000 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --no-overlap 2> $O/div10mvol_p$p.err ) > $O/div10mvol_p$p.json
  ( PG_VOL_PARTS=$p timeout 300 python bench.py --workload synthetic-vol --grid 2237 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --no-overlap 2> $O/syn10mvol_p$p.err ) > $O/syn10mvol_p$p.json
done
PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "vol or medium or fog" 2>&1 | tail -3
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05c/*vol_p*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms', {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, [ (k['kernel'][:10], round(k['avg_launch_ms'],2)) for k in d['roofline_kernels']])
    except Exception as e: print(f, 'ERR', e)
P
