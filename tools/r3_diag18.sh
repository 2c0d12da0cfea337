#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag18
rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "qwen_flash" > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 5 $O/pytest_a.log
timeout -k 10 200 python tools/bench_qwen_flash.py 48 > $O/bench48.log 2>&1 && echo bench48 ok
VQ3_QWEN_FAT_MAX=8 timeout -k 10 200 python tools/bench_qwen_flash.py 48 > $O/bench48_max8.log 2>&1 && echo bench48 max8 ok
grep -v amdgpu $O/bench48.log; grep -v amdgpu $O/bench48_max8.log | grep dense
A="--steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline"
timeout -k 10 400 python bench.py $A > $O/bench_fat.log 2> $O/e1 && echo b1 ok
VQ3_QWEN_FAT=0 timeout -k 10 400 python bench.py $A > $O/bench_nofat.log 2> $O/e2 && echo b2 ok
python - <<'PY'
import json
for f in ['bench_fat','bench_nofat']:
    d=json.loads(open('gpurun_out/r3_diag18/%s.log'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
PY
