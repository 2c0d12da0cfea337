#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag15
rm -rf $O; mkdir -p $O
for c in 6 7 13 17; do timeout -k 10 200 python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
