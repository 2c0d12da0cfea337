#!/bin/bash
# Tile-order sweep on the tower shapes (tools/bench_epilogue.py rows): VQ3_GEMM_XM x VQ3_GEMM_BAND against the traffic model's own choice.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_sweep
rm -rf $O; mkdir -p $O
timeout -k 10 120 python tools/bench_epilogue.py 30 49392 > $O/model.log 2>&1
for xm in 1 2 4 8; do for bw in 1 2 4 8 16; do
  VQ3_GEMM_XM=$xm VQ3_GEMM_BAND=$bw timeout -k 10 120 python tools/bench_epilogue.py 30 49392 > $O/xm${xm}_b${bw}.log 2>&1
  echo "xm $xm band $bw done"
done; done
python - <<'PY'
import glob, re, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/r3_sweep/*.log")):
    tag = f.split("/")[-1][:-4]
    for l in open(f):
        m = re.match(r"(.+?)\s+([\d.]+) us", l)
        if m: rows[m.group(1).strip()][tag] = float(m.group(2))
for name in ("fc1 +bias+gelu +ln_in", "qkv fused norm+rope +ln_in", "fc2 +bias+ls+res +st_out", "proj +bias+ls+res +st_out"):
    d = rows[name]
    best = sorted(d.items(), key=lambda kv: kv[1])[:4]
    print(f"{name:30s} model {d.get('model')}   best {best}")
PY
