#!/bin/bash
# round-4 small-kernel pass: the first block's backward with a tied-register load queue, fixed-order metrics / bias-gradient sums,
# 32 pieces per clip in the mel moments; then the stream-K row order.  Tests, per-kernel times, bench A/B against the same box.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_small
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_bf16_gpu.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1
tail -n 5 "$OUT/pytest.txt"
python tools/l0_time.py > "$OUT/l0_time.txt" 2>&1; cat "$OUT/l0_time.txt"
for i in 1 2; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench', d['value'], d['ms_per_step'], d['median_ms_per_step'], r['frac'], r['traffic'])" | tee -a "$OUT/bench.txt"
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/trace" -o b -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-overlap-dw --no-kernel-events > "$OLDPWD/$OUT/trace.log" 2>&1
cd "$OLDPWD"
python tools/trace_summary.py "$OUT/trace/b_kernel_trace.csv" 25 70 > "$OUT/trace_by_launch_shape.txt" 2>&1 || true
head -n 60 "$OUT/trace_by_launch_shape.txt" | cut -c1-150
rm -rf "$OUT/trace"
