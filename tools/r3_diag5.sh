#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag5
rm -rf $O; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1 && echo pytest gemm ok
for c in 20 22 30 -3; do python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
VQ3_GEMM_TABLE=1 python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err && echo bench ok
python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 && echo pytest all ok
tail -n 3 $O/pytest_all.log
