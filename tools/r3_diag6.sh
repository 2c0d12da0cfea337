#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag6
rm -rf $O; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1 && echo pytest gemm ok
python tools/gemm_stamps.py 49392 4096 1024 20 22 > $O/stamps_fc1.log 2>&1 && echo fc1 ok
python tools/gemm_stamps.py 49392 4096 1024 20 22 --epi=gelu > $O/stamps_fc1_gelu.log 2>&1 && echo fc1g ok
python tools/gemm_stamps.py 49392 1024 1024 20 22 --epi=res > $O/stamps_proj.log 2>&1 && echo proj ok
for c in 20 22 30 -3; do python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
python tools/bench_gemm_cfg.py 20 22 30 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,6144,2560 --shape=9600,2560,4096 --shape=6144,16384,4096 > $O/cfg.log 2>&1 && echo cfg ok
VQ3_GEMM_TABLE=1 python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err && echo bench ok
python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 && echo pytest all ok
tail -n 3 $O/pytest_all.log
