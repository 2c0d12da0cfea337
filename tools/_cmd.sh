cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b
mkdir -p $O
VQ3_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/serial -o s -- python3 bench.py --steps 8 --grad-accum 4 --warmup 4 --no-variants --no-trim-variant --no-cpu-baseline > $O/serial.log 2>&1 && python tools/trace_by_shape.py $O/serial/s_kernel_trace.csv $O/byshape.csv > $O/byshape.txt 2>&1; tail -3 $O/byshape.txt; rm -rf $O/serial
