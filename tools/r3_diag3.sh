#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag3
rm -rf $O; mkdir -p $O
python tools/gemm_stamps.py 49392 4096 1024 20 21 22 > $O/stamps_fc1.log 2>&1 && echo fc1 ok
python tools/gemm_stamps.py 49392 4096 1024 20 22 --epi=gelu > $O/stamps_fc1_gelu.log 2>&1 && echo fc1 gelu ok
python tools/gemm_stamps.py 49392 1024 4096 20 22 --epi=res > $O/stamps_fc2.log 2>&1 && echo fc2 ok
python tools/gemm_stamps.py 8192 8192 8192 20 > $O/stamps_sq.log 2>&1 && echo sq ok
python tools/gemm_stamps.py 9600 19456 2560 20 > $O/stamps_gu.log 2>&1 && echo gu ok
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err && echo bench ok
tail -c 3000 $O/bench.log
