#!/bin/bash
# Round-2 profiling passes (run on the GPU box through gpurun): kernel trace of the default bench, the serial accum-4 trace for the
# per-micro-batch breakdown, and four separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy / LDS conflicts).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_prof
mkdir -p $O
SHORT="bench.py --steps 8 --grad-accum 4 --warmup 4 --no-variants --no-trim-variant --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o d -- python3 bench.py --no-variants --no-trim-variant --no-cpu-baseline > $O/default.log 2>&1 && echo default ok
VQ3_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/serial -o s -- python3 $SHORT > $O/serial.log 2>&1 && echo serial ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $SHORT > $O/fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $SHORT > $O/write.log 2>&1 && echo write ok
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o m -- python3 $SHORT > $O/mfma.log 2>&1 && echo mfma ok
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -o l -- python3 $SHORT > $O/lds.log 2>&1 && echo lds ok
ls -la $O/*/ | head -40
