#!/bin/bash
# quick kernel-time breakdown of the step (no PMC): overlap on / off
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_now
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-mode"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o b -- $CMD --no-kernel-events > "$OUT/trace.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_noovl" -o b -- $CMD --no-overlap-dw --no-kernel-events > "$OUT/trace_noovl.log" 2>&1
python tools/trace_summary.py "$OUT/trace_noovl/b_kernel_trace.csv" 35 60 > "$OUT/trace_by_launch_shape.txt" 2>&1
python tools/stream_summary.py "$OUT/trace/b_kernel_trace.csv" 35 14 > "$OUT/streams.txt" 2>&1
for d in trace trace_noovl; do cp "$OUT/$d/b_kernel_stats.csv" "$OUT/${d}_kernel_stats.csv" 2>/dev/null; done
rm -rf "$OUT/trace" "$OUT/trace_noovl"
tail -2 "$OUT/trace.log"; tail -2 "$OUT/trace_noovl.log"
