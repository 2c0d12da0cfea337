#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag14
rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_trainer_gpu.py tests/test_fullwidth_gpu.py -x -q > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 3 $O/pytest_a.log
VQ3_GEMM_TABLE=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench.log 2> $O/bench.err && echo bench ok
VQ3_EMBED_BWD_LIVE=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench_off.log 2> $O/bench_off.err && echo bench off ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/prof.log 2>&1 && python tools/step_breakdown.py $O/p/p_kernel_trace.csv $O/breakdown.csv > /dev/null && echo prof ok
rm -rf $O/p
