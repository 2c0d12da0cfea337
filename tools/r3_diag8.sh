#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag8
rm -rf $O; mkdir -p $O
for st in 0 40 100 200 400; do VQ3_V6_STAGGER=$st python tools/bench_epilogue.py 30 49392 > $O/epi30_st$st.log 2>&1 && echo st $st ok; done
VQ3_V6_STAGGER=100 python tools/gemm_stamps.py 49392 1024 1024 20 --epi=res > $O/stamps_proj_st100.log 2>&1
VQ3_V6_STAGGER=100 python tools/bench_gemm_cfg.py 20 30 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,2560,4096 > $O/cfg_st100.log 2>&1
python tools/bench_gemm_cfg.py 20 30 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,2560,4096 > $O/cfg_st0.log 2>&1
