#!/bin/bash
# Register / spill summary of csrc/decode_layers.hip (run from anywhere): scratch instructions per 1000 asm lines.
set -e
D=/root/repo/vggt_qwen3_amd/csrc
mkdir -p /tmp/t
cd $D && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/root/repo/include -I. -mllvm -amdgpu-mfma-vgpr-form=1 "$@" --save-temps=obj -c decode_layers.hip -o /tmp/t/dl.o 2>&1 | grep -E "error" -A5 | head -30
S=/tmp/t/decode_layers-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count)" $S
grep -n "scratch_" $S | awk -F: '{print int($1/1000)*1000}' | uniq -c | tr '\n' ' '
echo
