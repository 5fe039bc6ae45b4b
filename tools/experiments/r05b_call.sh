mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
( timeout 900 python tools/fullsize_parity.py 41 51 "--fingerprint-in=tests/golden_large/fullframe_reference_fingerprint_config{config}.json" --out=$O/fullframe_parity.json > $O/fullframe_parity.log 2>&1 ); tail -3 $O/fullframe_parity.log | cut -c1-900
for v in 0 1; do
  ( PG_VOL_PARTS=$v timeout 600 python bench.py --workload divergent-vol --tris 10000000 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --no-overlap 2> $O/div10mvol_parts$v.err ) > $O/div10mvol_parts$v.json
  ( PG_VOL_PARTS=$v timeout 600 python bench.py --workload synthetic-vol --grid 2237 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --no-overlap 2> $O/syn10mvol_parts$v.err ) > $O/syn10mvol_parts$v.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/*parts*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms', {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, [ (k['kernel'][:10], round(k['avg_launch_ms'],2)) for k in d['roofline_kernels']])
    except Exception as e: print(f, 'ERR', e)
P
