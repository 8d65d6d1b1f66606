#!/bin/bash
# conv_taps_kernel EPI 2 (input gradient + normalisation-backward sums): y loads hoisted above the stores.  Tests, then the train step with
# the previous library and this one on one box (kernel trace + bench)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_epi2
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1; tail -n 2 "$OUT/pytest.txt"
bash tools/debug/r04_trace_ab.sh prev:speechdrivestemplates_amd/lib/libsdt_hip_prev.so cur:- > "$OUT/trace_ab.txt" 2>&1
grep "==\|total kernel\|conv_taps_kernel<64, 64, true" "$OUT/trace_ab.txt" | cut -c1-140
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
