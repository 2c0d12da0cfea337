#!/bin/bash
# Round-3 closing run on the GPU box: full GPU suite, smoke(), the driver's bench command with the per-shape GEMM table.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_final
rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 && echo pytest gpu ok
tail -n 3 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 && echo smoke ok
VQ3_GEMM_TABLE=1 timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err && echo bench ok
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_final/bench.log').read().strip().splitlines()[-1])
print('value',d['value'], d['ms_per_step'], 'roof',d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], 'fwd',d['forward_only']['forward_mfma_frac'], d['forward_only']['batched_8_micro_batches']['forward_mfma_frac'])
print('tg1',d['text_group_1_variant']['value'],'accum1',d['accum1_variant']['value'],'c4',d['c4_variant']['value'],'c5',d['c5_variant']['value'],'trim',d['trimmed_padding_variant']['value'])
PY
