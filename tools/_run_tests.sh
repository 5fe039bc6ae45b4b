OUT=gpurun_out/${1:-tests}; mkdir -p $OUT; export TMPDIR=/tmp
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
