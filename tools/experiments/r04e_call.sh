OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s); python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench.py wall $(( $(date +%s) - T0 )) s"; grep "bench:" $OUT/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/bench_default.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'])
print({k:r.get(k) for k in ('kernel','frac','traffic','bound')})
print('traffic_source', r.get('traffic_source'))
print('hbm_side', r.get('hbm_side'))
print('gather', {k:r['gather'].get(k) for k in ('frac','l2_hit_rate','l2_hit_rate_source')} if r.get('gather') else None)
print('hbm_regime', {k:(v if not isinstance(v,dict) else v) for k,v in r.get('hbm_regime',{}).items() if k in ('frac','hbm_side','traffic','traffic_source')})
print('vector_issue', r.get('vector_issue'))
print('cpu', json.dumps(d.get('cpu_baseline'))[:600])
PY
