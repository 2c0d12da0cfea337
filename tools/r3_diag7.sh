#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag7
rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_vggt_gpu.py -x -q -k "gemm or qkv" > $O/pytest_gemm.log 2>&1 && echo pytest gemm ok
tail -n 30 $O/pytest_gemm.log
timeout -k 10 200 python tools/bench_gemm_cfg.py 20 22 23 30 --shape=49392,4096,1024 --shape=49392,3072,1024 --shape=49392,1024,1024 --shape=49392,1024,4096 --shape=9600,6144,2560 > $O/cfg.log 2>&1 && echo cfg ok
for c in 23 30; do timeout -k 10 200 python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
