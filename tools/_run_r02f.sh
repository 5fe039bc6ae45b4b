OUT=gpurun_out/r02f; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sampler_ or 02sequence" 2>&1 | tail -25 ) > $OUT/pytest_samplers.log
tail -12 $OUT/pytest_samplers.log
( PBRT_SKIP_SLOW=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
