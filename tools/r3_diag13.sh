#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag13
rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "tune_file or rmsnorm" > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 3 $O/pytest_a.log
VQ3_GEMM_AUTOTUNE_LOG=1 VQ3_GEMM_TABLE=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench.log 2> $O/bench.err && echo bench ok
VQ3_GEMM_PREFER_SPLIT=1 VQ3_GEMM_TABLE=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench_split.log 2> $O/bench_split.err && echo bench split ok
bash tools/r3_flash_pmc.sh
