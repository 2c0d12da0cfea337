#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag21
rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "qwen_flash" > $O/pytest_a.log 2>&1 && echo pytest a ok
tail -n 3 $O/pytest_a.log
VQ3_QWEN_DKV_LDS=1 timeout -k 10 200 python tools/diag/qwen_dkv_dump.py /tmp/dkv_new.pt > $O/dump_new.log 2>&1 && echo dump new ok
VQ3_QWEN_DKV_LDS=0 timeout -k 10 200 python tools/diag/qwen_dkv_dump.py /tmp/dkv_old.pt > $O/dump_old.log 2>&1 && echo dump old ok
python - > $O/compare.log 2>&1 <<'PY'
import torch
a, b = torch.load("/tmp/dkv_new.pt"), torch.load("/tmp/dkv_old.pt")
bad = 0
for k in a:
    for n, x, y in zip(("dQ", "dK", "dV"), a[k], b[k]):
        if not torch.equal(x, y):
            bad += 1
            print(k, n, "max abs diff", (x.float() - y.float()).abs().max().item())
print("mismatches:", bad, "of", 3 * len(a))
PY
tail -3 $O/compare.log
true
true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/prof.log 2>&1 && python tools/step_breakdown.py $O/p/p_kernel_trace.csv $O/breakdown.csv > /dev/null && echo prof ok
rm -rf $O/p
grep -E "qwen_flash|qkprep|TOTAL|window" $O/breakdown.csv
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/bench.log 2> $O/e1 && python -c "
import json
d=json.loads(open('$O/bench.log').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
