cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04m
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04m/trace -o b -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-mode --no-kernel-events --no-overlap-dw > gpurun_out/r04m/trace.log 2>&1
python tools/trace_summary.py gpurun_out/r04m/trace/b_kernel_trace.csv 14 70 > gpurun_out/r04m/by_shape.txt 2>&1
rm -rf gpurun_out/r04m/trace
grep -n "false, 0, 0\|splitk_reduce\|total kernel" gpurun_out/r04m/by_shape.txt
