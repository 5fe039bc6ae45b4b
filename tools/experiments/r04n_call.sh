# r04n: which build first differs from the reference binary on the config-4 window (76 last-bit pixels in r04z's slow test): the
# library of commit 3ab88b9 (base) and the current one, each with materials evaluated ahead (pre=1) and inside k_shade<2> (pre=0).
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
for v in base default; do for pre in 1 0; do
  L=$PWD/pbrt-v3_amd/libpbrt_gpu.so; [ $v != default ] && L=$PWD/gpurun_in_libpbrt_gpu_$v.so
  PG_ANYHIT_ORDER=reference PG_MAT_PRE=$pre PBRT_GPU_LIB=$L timeout 300 python tools/fullsize_parity.py 4 --out=$OUT/cfg4_${v}_pre$pre.json > $OUT/cfg4_${v}_pre$pre.log 2>&1
  python -c "
import json; d=json.load(open('$OUT/cfg4_${v}_pre$pre.json')); d=d[0] if isinstance(d,list) else d
print('$v pre=$pre', d['pixels_differing'], d['max_rel_err'], d['device_render_ms'])"
done; done
