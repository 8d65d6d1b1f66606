#!/bin/bash
# time-sliced issue priority between the two workgroups of a CU (convsk_kernel, SK_FAIR): per-workgroup finish times, per-layer rates and the
# train step against the previous library (lib/libsdt_hip_prev.so, built by hand from `git archive HEAD`) on one box
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_fair
mkdir -p "$OUT"
PREV=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_prev.so
python tools/debug/sk_timeline.py --dtype f32 --only L2,L4 --roles fwd,dX 2>&1 | grep -v amdgpu.ids > "$OUT/sk_timeline.txt"
python - <<'PY' >> "$OUT/sk_timeline.txt"
import numpy as np
for name in ("L2_fwd", "L4_fwd", "L4_dX"):
    tl = np.load("gpurun_out/sk_tl_%s.npy" % name)
    used = tl[:, :, 0] != 0
    t = tl[..., :5].astype(np.float64) * 0.01
    t0 = t[..., 0][used].min()
    r = np.arange(512); bid = (r % 64) * 8 + r // 64
    end = np.array([t[i, :, 4][used[i]].max() for i in range(512)]) - t0
    steps = np.array([tl[i, :, 5][used[i]].sum() for i in range(512)])
    kl = np.array([(t[i, :, 3] - t[i, :, 2])[used[i]].sum() for i in range(512)]) / steps
    lo = bid < 256
    print("%s: first-half workgroups finish at %.1f us (K step %.3f us), second-half at %.1f us (K step %.3f us); launch %.1f us"
          % (name, np.median(end[lo]), np.median(kl[lo]), np.median(end[~lo]), np.median(kl[~lo]), end.max()))
PY
cat "$OUT/sk_timeline.txt" | tail -n 4
for i in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export SDT_HIP_LIB=$PREV; else unset SDT_HIP_LIB; fi
    echo "== $lib" >> "$OUT/conv_bench.txt"
    python tools/conv_bench.py --roles fwd,dX,dW --streamk 2 --only L2,L3,L4,L5,L6,L7 2>&1 | grep -v amdgpu.ids >> "$OUT/conv_bench.txt"
  done
done
cat "$OUT/conv_bench.txt" | tail -n 52
for i in 1 2 3; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export SDT_HIP_LIB=$PREV; else unset SDT_HIP_LIB; fi
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$lib', round(d['value'],1), round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(r['frac'],4), round(r['avg_launch_us'],1), d['streamk_errors'])" | tee -a "$OUT/bench.txt"
  done
done
