#!/bin/bash
# Round-5 profiling passes (run on the GPU box through gpurun; usage: bash tools/run_profiles_r5.sh <commit> [c2|c4|all]): kernel stats of the default
# bench, the serial accum-8 trace for the per-micro-batch / per-shape breakdowns, and four separate --pmc passes (FETCH_SIZE / WRITE_SIZE /
# MFMA busy / LDS conflicts). The summaries land in gpurun_out/r5_prof/out (copy them to profiles/); the raw traces are deleted.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMIT=${1:-unknown}
PART=${2:-all}
O=gpurun_out/r5_prof_$PART
rm -rf $O; mkdir -p $O/out
SHORT="bench.py --steps 20 --warmup 20 --no-variants --no-trim-variant --no-cpu-baseline"     # the driver's schedule: one window of 20 = two passes of 10
if [ "$PART" != c4 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -o d -- python3 bench.py --steps 20 --warmup 5 --no-variants --no-trim-variant --no-cpu-baseline > $O/default.log 2>&1 && cp $O/default/d_kernel_stats.csv $O/out/r5_driver_cmd_kernel_stats.csv && echo driver-cmd ok
rm -rf $O/default
VQ3_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/serial -o s -- python3 $SHORT > $O/serial.log 2>&1 && python tools/step_breakdown.py $O/serial/s_kernel_trace.csv $O/out/r5_step_breakdown_accum20_serial.csv > /dev/null && python tools/trace_by_shape.py $O/serial/s_kernel_trace.csv $O/out/r5_step_by_shape_accum20_serial.csv > /dev/null && echo serial ok
rm -rf $O/serial
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $SHORT > $O/fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $SHORT > $O/write.log 2>&1 && echo write ok
python tools/pmc_step_by_shape.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/out/r5_gemm_fetch_by_shape.txt > /dev/null && echo by-shape traffic ok
python tools/summarize_pmc.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/out/r5_pmc "$COMMIT" "rocprofv3 --pmc <counter> -- python3 $SHORT (window: the longest accumulation window of the run)" > /dev/null && echo traffic ok
rm -rf $O/fetch $O/write
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o m -- python3 $SHORT > $O/mfma.log 2>&1 && echo mfma ok
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -o l -- python3 $SHORT > $O/lds.log 2>&1 && echo lds ok
python tools/summarize_pmc_mfma.py $O/mfma/m_counter_collection.csv $O/lds/l_counter_collection.csv $O/out/r5_pmc_mfma_lds.csv > /dev/null && echo mfma_lds ok
rm -rf $O/mfma $O/lds
fi
if [ "$PART" != c2 ]; then
# ---- config C4 (8 views + geometry): kernel stats, by shape, MFMA busy, traffic
C4="bench.py --views 8 --geom --steps 8 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o c -- python3 $C4 > $O/c4.log 2>&1 && cp $O/c4/c_kernel_stats.csv $O/out/r5_c4_kernel_stats.csv && python tools/trace_by_shape.py $O/c4/c_kernel_trace.csv $O/out/r5_c4_by_shape.csv > /dev/null && echo c4 ok
rm -rf $O/c4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c4m -o m -- python3 $C4 > $O/c4m.log 2>&1 && echo c4 mfma ok
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c4l -o l -- python3 $C4 > $O/c4l.log 2>&1 && echo c4 lds ok
python tools/summarize_pmc_mfma.py $O/c4m/m_counter_collection.csv $O/c4l/l_counter_collection.csv $O/out/r5_c4_pmc_mfma_lds.csv > /dev/null && echo c4 mfma_lds ok
rm -rf $O/c4m $O/c4l
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4f -o f -- python3 $C4 > $O/c4f.log 2>&1 && echo c4 fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4w -o w -- python3 $C4 > $O/c4w.log 2>&1 && echo c4 write ok
python tools/summarize_pmc.py $O/c4f/f_counter_collection.csv $O/c4w/w_counter_collection.csv $O/out/r5_c4_pmc "$COMMIT" "rocprofv3 --pmc <counter> -- python3 $C4" > /dev/null && echo c4 traffic ok
rm -rf $O/c4f $O/c4w
fi
# decode: the persistent layer-stack kernel against the per-projection launches, kernel stats, bytes per launch
python tools/bench_decode.py --new 64 2>/dev/null | tail -1 > $O/out/r5_decode_bench.jsonl && VQ3_DECODE_PERSISTENT=0 python tools/bench_decode.py --new 64 2>/dev/null | tail -1 >> $O/out/r5_decode_bench.jsonl && echo decode bench ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -o dec -- python3 tools/bench_decode.py --new 64 > $O/dec.log 2>&1 && cp $O/dec/dec_kernel_stats.csv $O/out/r5_decode_kernel_stats.csv && echo decode stats ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/decf -o f -- python3 tools/bench_decode.py --new 16 > $O/decf.log 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/decw -o w -- python3 tools/bench_decode.py --new 16 > $O/decw.log 2>&1 && python tools/summarize_decode_pmc.py $O/decf/f_counter_collection.csv $O/decw/w_counter_collection.csv $O/out/r5_decode_pmc_traffic.json > /dev/null && echo decode traffic ok
rm -rf $O/dec $O/decf $O/decw
# (r5_decode_layer_stamps.txt needs the diagnostic build: make -C vggt_qwen3_amd/csrc EXTRA=-DVQ3_DL_STAMPS after touching decode_layers.hip,
#  then python tools/diag/decode_layers_stamps.py 36)
# flash attention (VGGT): the two workgroup placements, HIP-event micro-benchmark (tools/bench_flash.py)
VQ3_FLASH_XCD=0 python tools/bench_flash.py 2>/dev/null | sed 's/^/xcd=0 /' > $O/out/r5_flash_bench.txt; python tools/bench_flash.py 2>/dev/null | sed 's/^/xcd=1 /' >> $O/out/r5_flash_bench.txt; echo flash bench ok
# the full-depth parity report (tests/test_fulldepth_gpu.py writes gpurun_out/r5_depth_parity.json)
python -m pytest tests/test_fulldepth_gpu.py -q -m gpu > $O/fulldepth.log 2>&1 && cp gpurun_out/r5_depth_parity.json $O/out/r5_depth_parity.json && echo depth parity ok
ls -la $O/out
