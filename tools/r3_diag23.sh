#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag23
rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "rmsnorm" > $O/pytest_a.log 2>&1 && echo pytest ok
tail -n 2 $O/pytest_a.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/prof.log 2>&1 && python tools/step_breakdown.py $O/p/p_kernel_trace.csv $O/breakdown.csv > /dev/null && echo prof ok
rm -rf $O/p
grep -E "rmsnorm|colsum|TOTAL|window" $O/breakdown.csv
