#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag4
rm -rf $O; mkdir -p $O
SH="--shape=49392,4096,1024 --shape=49392,3072,1024 --shape=49392,1024,1024 --shape=49392,1024,4096 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,6144,2560 --shape=9600,2560,4096 --shape=6144,16384,4096"
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1 && echo pytest gemm ok
python tools/bench_gemm_cfg.py 6 7 17 20 22 30 $SH > $O/cfg.log 2>&1 && echo cfg ok
VQ3_V6_LEAN=0 python tools/bench_gemm_cfg.py 20 22 30 $SH > $O/cfg_nolean.log 2>&1 && echo nolean ok
python tools/gemm_stamps.py 49392 4096 1024 20 22 > $O/stamps_fc1.log 2>&1 && echo fc1 ok
python tools/gemm_stamps.py 9600 19456 2560 20 > $O/stamps_gu.log 2>&1 && echo gu ok
python tools/gemm_stamps.py 9600 2560 9728 20 --epi=res > $O/stamps_down.log 2>&1 && echo down ok
for c in 17 20 22; do python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
