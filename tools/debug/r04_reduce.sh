#!/bin/bash
# stream-K weight-gradient slab reduce with Q lanes per float4: tests, the as-run kernel sequence of one step, A/B against the previous library
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_reduce
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1; tail -n 2 "$OUT/pytest.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/trace" -o b -- python "$ROOT/bench.py" --steps 12 --warmup 5 --no-cpu-baseline --no-alt-mode --no-kernel-events > "$ROOT/$OUT/trace.log" 2>&1)
python tools/debug/step_sequence.py "$OUT/trace/b_kernel_trace.csv" 3 > "$OUT/step_sequence.txt" 2>&1
rm -rf "$OUT/trace"
bash tools/debug/r04_trace_ab.sh prev:speechdrivestemplates_amd/lib/libsdt_hip_prev.so cur:- > "$OUT/trace_ab.txt" 2>&1
grep "==\|total kernel\|dw_sk_reduce" "$OUT/trace_ab.txt" | cut -c1-140
PREV=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_prev.so
for i in 1 2 3; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export SDT_HIP_LIB=$PREV; else unset SDT_HIP_LIB; fi
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$lib', round(d['value'],1), round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(r['frac'],4), round(r['avg_launch_us'],1), d['streamk_errors'])" | tee -a "$OUT/bench.txt"
  done
done
