#!/bin/bash
# Regenerates vggt_qwen3_amd/gemm_tune_gfx950.txt, the kernel-choice table the library loads by default (gemm.hip: tuned_choice).
# Run on an MI355X box from the repo root:   bash tools/make_tune_table.sh   (writes gpurun_out/gemm_tune_gfx950.txt; copy it into the package)
# Every GEMM shape the Stage-1 step meets - config C2 at passes of 10 / 8 / 1, accumulation 1, the trimmed-padding and forward-only
# windows, configs C4 / C5 - is measured once (5 cold launches per candidate kernel) by the bench itself with the shipped table off.
set -e
mkdir -p gpurun_out
T=gpurun_out/tune_raw.txt
rm -f $T
export VQ3_GEMM_TUNE_TABLE=0 VQ3_GEMM_TUNE_FILE=$T
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tune_bench20.json 2> gpurun_out/tune_bench20.err
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/tune_bench8.json 2> gpurun_out/tune_bench8.err
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-variants --train-projector > gpurun_out/tune_bench8p.json 2> gpurun_out/tune_bench8p.err
{
  echo "# vq3 GEMM kernel choices for gfx950 (MI355X): M N K batch flags cfg - measured by tools/make_tune_table.sh ${1:+at commit $1}"
  echo "# keys: gemm.hip tuned_choice (M < 256 in steps of 32, K < 512 in steps of 64; flags = layout / epilogue kind / alignment bits)"
  sort -u $T | sort -n -k1,1 -k2,2 -k3,3 -k5,5
} > gpurun_out/gemm_tune_gfx950.txt
wc -l gpurun_out/gemm_tune_gfx950.txt
