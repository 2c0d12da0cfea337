#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_flash
mkdir -p $O
for v in 0 1; do
VQ3_FLASH_VAR=$v timeout -k 10 300 python -m pytest tests/test_vggt_gpu.py tests/test_fullwidth_gpu.py -x -q > $O/pytest_$v.log 2>&1; echo "var $v pytest rc=$?"; tail -n 2 $O/pytest_$v.log
VQ3_FLASH_VAR=$v timeout -k 10 200 python tools/bench_flash.py > $O/bench_$v.log 2>&1; cat $O/bench_$v.log | grep -v amdgpu
done
