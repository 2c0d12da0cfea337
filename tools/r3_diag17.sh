#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag17
rm -rf $O; mkdir -p $O
A="--steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline"
timeout -k 10 400 python bench.py $A > $O/bench_new_auto.log 2> $O/e1 && echo b1 ok
VQ3_QWEN_KV_PARTS=1 timeout -k 10 400 python bench.py $A > $O/bench_new_p1.log 2> $O/e2 && echo b2 ok
VQ3_QWEN_DKV_LDS=0 timeout -k 10 400 python bench.py $A > $O/bench_old_auto.log 2> $O/e3 && echo b3 ok
python - <<'PY'
import json
for f in ['bench_new_auto','bench_new_p1','bench_old_auto']:
    d=json.loads(open('gpurun_out/r3_diag17/%s.log'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
PY
