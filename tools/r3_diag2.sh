#!/bin/bash
# Round-3 diagnostics 2: banded tile order A/B (VQ3_GEMM_BAND: 0 = round-2 m-fastest walk, unset = traffic model, n = forced width)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_diag2
rm -rf $O; mkdir -p $O
SH="--shape=49392,4096,1024 --shape=49392,3072,1024 --shape=49392,1024,1024 --shape=49392,1024,4096 --shape=9600,19456,2560 --shape=9600,2560,9728 --shape=9600,6144,2560 --shape=9600,2560,4096"
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1 && echo pytest gemm ok
for b in 0 model 2 4 8 16; do
  if [ $b = model ]; then unset VQ3_GEMM_BAND; else export VQ3_GEMM_BAND=$b; fi
  python tools/bench_gemm_cfg.py 7 11 20 21 22 30 $SH > $O/cfg_band_$b.log 2>&1 && echo band $b ok
done
unset VQ3_GEMM_BAND
for x in 1 2 4 8; do
  VQ3_GEMM_XM=$x python tools/bench_gemm_cfg.py 20 22 $SH > $O/cfg_xm_$x.log 2>&1 && echo xm $x ok
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/write.log 2>&1 && echo write ok
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -o h -- python3 tools/bench_gemm_cfg.py 7 11 20 21 22 $SH > $O/hit.log 2>&1 && echo hit ok
python tools/pmc_by_shape.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/hit/h_counter_collection.csv > $O/pmc_by_shape.txt 2>&1
rm -rf $O/fetch $O/write $O/hit
python -m pytest tests/test_trainer_gpu.py tests/test_kernels_gpu.py -x -q > $O/pytest_rest.log 2>&1 && echo pytest rest ok
