#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_tests
rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_perceiver_bwd_gpu.py -x -q > $O/perceiver.log 2>&1; echo "perceiver rc=$?"
tail -n 25 $O/perceiver.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_perceiver_bwd_gpu.py > $O/all.log 2>&1; echo "all rc=$?"
tail -n 25 $O/all.log
