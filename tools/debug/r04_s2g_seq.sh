#!/bin/bash
# ordered kernel sequence of one s2g train step (gaps = host-bound stretches).   gpurun -- bash tools/debug/r04_s2g_seq.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04k
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04k/trace -o b -- python bench.py --config voice2pose_s2g --steps 12 --warmup 4 --no-cpu-baseline --no-alt-mode --no-kernel-events > gpurun_out/r04k/trace2.log 2>&1
python tools/debug/step_sequence.py gpurun_out/r04k/trace/b_kernel_trace.csv 3 > gpurun_out/r04k/s2g_sequence.txt 2>&1
python tools/stream_summary.py gpurun_out/r04k/trace/b_kernel_trace.csv 16 10 > gpurun_out/r04k/s2g_streams.txt 2>&1
rm -rf gpurun_out/r04k/trace
python tools/host_time.py --config voice2pose_s2g > gpurun_out/r04k/s2g_host_time.txt 2>&1
tail -n 3 gpurun_out/r04k/s2g_host_time.txt
