#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_flash
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_vggt_gpu.py tests/test_fullwidth_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
timeout -k 10 200 python tools/bench_flash.py > $O/bench.log 2>&1; cat $O/bench.log | grep -v amdgpu
