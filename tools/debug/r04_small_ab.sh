#!/bin/bash
# same-box A/B of the small-kernel pass + stream-K row order: the previous commit's library (lib/libsdt_hip_prev.so, built by hand from
# `git archive HEAD`) against the current one, alternating
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_small_ab
mkdir -p "$OUT"
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
for lib in prev cur; do
  if [ $lib = prev ]; then export SDT_HIP_LIB=$PREV; else unset SDT_HIP_LIB; fi
  echo "== $lib" >> "$OUT/l0_time.txt"; python tools/l0_time.py >> "$OUT/l0_time.txt" 2>&1
done
cat "$OUT/l0_time.txt"
