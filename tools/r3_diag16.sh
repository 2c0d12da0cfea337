#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag16
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
        eq = torch.equal(x, y)
        if not eq:
            bad += 1
            print(k, n, "max abs diff", (x.float() - y.float()).abs().max().item())
print("mismatches:", bad, "of", 3 * len(a))
PY
cat $O/compare.log | tail -5
VQ3_QWEN_DKV_LDS=1 timeout -k 10 200 python tools/bench_qwen_flash.py 48 > $O/bench_new.log 2>&1 && echo bench new ok
VQ3_QWEN_DKV_LDS=0 timeout -k 10 200 python tools/bench_qwen_flash.py 48 > $O/bench_old.log 2>&1 && echo bench old ok
grep -v amdgpu $O/bench_new.log; grep -v amdgpu $O/bench_old.log
