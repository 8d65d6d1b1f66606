#!/bin/bash
# per-kernel time of the s2g (BatchNorm + LSGAN) config.   gpurun -- bash tools/debug/r04_s2g_trace.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04k
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04k/trace -o b -- python bench.py --config voice2pose_s2g --steps 12 --warmup 4 --no-cpu-baseline --no-alt-mode --no-kernel-events --no-overlap-dw > gpurun_out/r04k/trace.log 2>&1
python tools/trace_summary.py gpurun_out/r04k/trace/b_kernel_trace.csv 12 45 > gpurun_out/r04k/s2g_by_launch_shape.txt 2>&1
rm -rf gpurun_out/r04k/trace
head -n 50 gpurun_out/r04k/s2g_by_launch_shape.txt
