#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag22
rm -rf $O; mkdir -p $O
VQ3_QWEN_FAT=2 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "qwen_flash" > $O/pytest_a.log 2>&1 && echo pytest fat2 ok
tail -n 3 $O/pytest_a.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "qwen_flash" > $O/pytest_b.log 2>&1 && echo pytest fat1 ok
VQ3_QWEN_FAT=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/prof.log 2>&1 && python tools/step_breakdown.py $O/p/p_kernel_trace.csv $O/breakdown.csv > /dev/null && echo prof ok
rm -rf $O/p
grep -E "qwen_flash|TOTAL|window" $O/breakdown.csv
A="--steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline"
VQ3_QWEN_FAT=2 timeout -k 10 400 python bench.py $A > $O/bench_fat2.log 2> $O/e1 && echo b1 ok
timeout -k 10 400 python bench.py $A > $O/bench_def.log 2> $O/e2 && echo b2 ok
python - <<'PY'
import json
for f in ['bench_fat2','bench_def']:
    d=json.loads(open('gpurun_out/r3_diag22/%s.log'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
PY
