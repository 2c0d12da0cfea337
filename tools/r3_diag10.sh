#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag10
rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or linear" > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 3 $O/pytest_a.log
for c in 20 30; do timeout -k 10 200 python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
timeout -k 10 200 python tools/gemm_stamps.py 49392 1024 1024 20 22 --epi=res > $O/stamps_proj.log 2>&1 && echo proj ok
timeout -k 10 200 python tools/bench_gemm_cfg.py 20 22 30 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,6144,2560 --shape=9600,2560,4096 --shape=6144,16384,4096 > $O/cfg.log 2>&1 && echo cfg ok
VQ3_GEMM_TABLE=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench.log 2> $O/bench.err && echo bench ok
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_vlm_gpu.py tests/test_trainer_gpu.py -x -q > $O/pytest_b.log 2>&1 && echo pytest b ok
tail -n 3 $O/pytest_b.log
