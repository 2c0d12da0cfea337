#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag12
rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -x -q -k "rmsnorm or abi or symbols" > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 3 $O/pytest_a.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench.log 2> $O/bench.err && echo bench ok
bash tools/run_profiles_r3.sh 8635650 c4
