#!/bin/bash
# Register / spill summary of csrc/decode_layers.hip: VGPR count, spilled registers and scratch instructions per 1000 lines of its ISA
# (the persistent decode kernel must need NO scratch - tests/test_abi.py checks the linked object). Extra hipcc flags pass through,
# e.g. tools/diag/dl_regs.sh -DVQ3_DL_RING=24
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/dl_regs
mkdir -p "$OUT"
cd "$ROOT/vggt_qwen3_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$ROOT/include" -I. -mllvm -amdgpu-mfma-vgpr-form=1 "$@" --save-temps=obj \
  -c decode_layers.hip -o "$OUT/dl.o" 2>&1 | grep -E "error" -A5 | head -30 || true
S="$OUT/decode_layers-hip-amdgcn-amd-amdhsa-gfx950.s"
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size)" "$S"
grep -n "scratch_" "$S" | awk -F: '{print int($1/1000)*1000}' | uniq -c | tr '\n' ' '
echo
