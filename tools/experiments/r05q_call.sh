#!/bin/bash
# The whole GPU suite after the moving shapes / instances went in (the binding rebuilt against ABI 26), with the slow full-size comparisons.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export TMPDIR=/tmp
OUT=gpurun_out/r05q; mkdir -p $OUT
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -45 ) > $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log
