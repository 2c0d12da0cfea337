cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fwd; mkdir -p gpurun_out/fwd
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fwd/t -o f -- python3 tools/prof_forward.py 16 > gpurun_out/fwd/run.log 2>&1 && python tools/prof_forward.py --summarize $(find gpurun_out/fwd/t -name f_kernel_trace.csv) 16 gpurun_out/fwd/r5_forward_b6_by_shape.csv > gpurun_out/fwd/sum.log 2>&1; rm -rf gpurun_out/fwd/t; tail -3 gpurun_out/fwd/run.log; head -50 gpurun_out/fwd/sum.log
