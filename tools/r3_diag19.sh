#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag19
rm -rf $O; mkdir -p $O
A="--steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline"
timeout -k 10 400 python bench.py $A > $O/bench_fat4.log 2> $O/e1 && echo b1 ok
VQ3_QWEN_FAT=0 timeout -k 10 400 python bench.py $A > $O/bench_nofat.log 2> $O/e2 && echo b2 ok
VQ3_QWEN_FAT_HOSTREAD=1 timeout -k 10 400 python bench.py $A > $O/bench_fathost.log 2> $O/e3 && echo b3 ok
timeout -k 10 400 python bench.py $A > $O/bench_fat4b.log 2> $O/e4 && echo b4 ok
python - <<'PY'
import json
for f in ['bench_fat4','bench_nofat','bench_fathost','bench_fat4b']:
    d=json.loads(open('gpurun_out/r3_diag19/%s.log'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/prof.log 2>&1 && python tools/step_breakdown.py $O/p/p_kernel_trace.csv $O/breakdown.csv > /dev/null && echo prof ok
rm -rf $O/p
grep -E "qwen_flash|qkprep|TOTAL|window" $O/breakdown.csv
