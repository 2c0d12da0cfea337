#!/bin/bash
# Round-3 diagnostics: per-config table of the VGGT-tower GEMM shapes at M = 49392 (cold weights), epilogue costs, and a FETCH_SIZE pass.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag1
rm -rf $O; mkdir -p $O
SH="--shape=49392,4096,1024 --shape=49392,3072,1024 --shape=49392,1024,1024 --shape=49392,1024,4096"
python tools/bench_gemm_cfg.py 7 9 11 13 20 21 22 30 $SH > $O/cfg.log 2>&1 && echo cfg ok
VQ3_V6_PERSIST=0 python tools/bench_gemm_cfg.py 21 22 $SH > $O/cfg_nopersist.log 2>&1 && echo nopersist ok
for c in -3 20 21 22; do python tools/bench_epilogue.py $c 49392 > $O/epi_$c.log 2>&1 && echo epi $c ok; done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/write.log 2>&1 && echo write ok
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -o h -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/hit.log 2>&1 && echo hit ok
ls -la $O $O/fetch $O/write $O/hit
