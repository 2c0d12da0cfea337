#!/bin/bash
# Tile-order sweep on the wide-N text GEMMs (tools/bench_wide_gemm.py): VQ3_GEMM_XM x VQ3_GEMM_BAND against the traffic model's own choice.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_wide_sweep
rm -rf $O; mkdir -p $O
timeout -k 10 120 python tools/bench_wide_gemm.py > $O/model.log 2>&1 || exit 1
for xm in 1 2 4 8; do for bw in 1 2 4 8; do
  VQ3_GEMM_XM=$xm VQ3_GEMM_BAND=$bw timeout -k 10 120 python tools/bench_wide_gemm.py > $O/xm${xm}_b${bw}.log 2>&1 || exit 1
  echo "xm $xm band $bw done"
done; done
for f in $O/*.log; do echo "== $f"; cat $f; done > $O/all.txt
