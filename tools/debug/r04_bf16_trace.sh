#!/bin/bash
# Kernel trace of the bf16-storage step, every kernel alone on the GPU (weight gradients on the main stream) and as run:  bash tools/debug/r04_bf16_trace.sh <outdir>
set -u
OUT=${1:-gpurun_out/r04_bf16}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python bench.py --storage bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-kernel-events"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_noovl" -o b -- $CMD --no-overlap-dw > "$OUT/trace_noovl.log" 2>&1
python tools/trace_summary.py "$OUT/trace_noovl/b_kernel_trace.csv" 25 70 > "$OUT/trace_by_launch_shape.txt" 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o b -- $CMD > "$OUT/trace.log" 2>&1
python tools/stream_summary.py "$OUT/trace/b_kernel_trace.csv" 25 16 > "$OUT/streams.txt" 2>&1
for d in trace trace_noovl; do cp "$OUT/$d/b_kernel_stats.csv" "$OUT/${d}_kernel_stats.csv" 2>/dev/null; done
rm -rf "$OUT/trace" "$OUT/trace_noovl"
head -45 "$OUT/trace_by_launch_shape.txt"
