# r05f: the animated-camera goldens and the binding on the device; the two divergent stand-ins with this run's own PMC passes
ulimit -c 0
mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
( PBRT_SKIP_SLOW=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_binding.py tests/test_gpu_anyhit_order.py -m gpu -q -k "camanim" 2>&1 | tail -4 ) > $O/pytest_camanim.log; cat $O/pytest_camanim.log
( timeout 600 python bench.py --workload divergent --tris 5000000 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/bench_div5m.err ) > $O/bench_div5m.json
( timeout 600 python bench.py --workload divergent-vol --tris 10000000 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/bench_div10m_vol.err ) > $O/bench_div10m_vol.json
python - <<'P'
import json
for f in ("gpurun_out/r05f/bench_div5m.json", "gpurun_out/r05f/bench_div10m_vol.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 1), 'ms', {k: round(v, 1) for k, v in d['kernel_ms_per_step'].items()})
        for k in d['roofline_kernels']:
            print('   ', k['kernel'][:12], 'ms/launch', round(k['avg_launch_ms'], 2), 'alg GB', round(k['algorithmic_bytes_per_launch'] / 1e9, 2), 'traffic GB', round((k.get('traffic') or 0) / 1e9, 2), 'issue', round(k.get('vector_issue', {}).get('frac', 0), 3))
    except Exception as e: print(f, 'ERR', e)
P
tail -2 $O/bench_div5m.err
