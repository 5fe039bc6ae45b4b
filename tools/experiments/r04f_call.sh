OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hlbvh_build.py -m gpu -x -q 2>&1 | tail -4 ) | tee $OUT/pytest_hlbvh.log
( timeout 600 python tools/hlbvh_timing.py 2>&1 | tail -4 ) | tee $OUT/hlbvh_timing.json
