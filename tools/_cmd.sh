cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b
mkdir -p $O
VQ3_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/serial -o s -- python3 bench.py --steps 16 --grad-accum 8 --warmup 8 --no-variants --no-trim-variant --no-cpu-baseline > $O/serial.log 2>&1 && python tools/trace_by_shape.py $O/serial/s_kernel_trace.csv $O/byshape_defer.csv > $O/byshape_defer.txt 2>&1; tail -3 $O/byshape_defer.txt; rm -rf $O/serial
