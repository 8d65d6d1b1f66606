#!/bin/bash
# per-kernel durations of the train step with two libraries on one box (rocprofv3 kernel trace, every kernel alone): $1 = label:path pairs
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_trace_ab
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
for pair in "$@"; do
  lab=${pair%%:*}; lib=${pair#*:}
  if [ "$lib" = "-" ]; then unset SDT_HIP_LIB; else export SDT_HIP_LIB=$ROOT/$lib; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/trace_$lab" -o b -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-overlap-dw --no-kernel-events > "$ROOT/$OUT/trace_$lab.log" 2>&1)
  python tools/trace_summary.py "$OUT/trace_$lab/b_kernel_trace.csv" 25 80 > "$OUT/by_launch_shape_$lab.txt" 2>&1
  rm -rf "$OUT/trace_$lab"
  echo "== $lab"; head -n 24 "$OUT/by_launch_shape_$lab.txt" | cut -c1-140
done
