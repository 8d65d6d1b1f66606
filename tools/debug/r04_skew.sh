#!/bin/bash
# stream-K ranges skewed towards the first-dispatched workgroups (sk_bound, SDT_SK_SKEW per mille in the tuning build): correctness, then rates
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_skew
mkdir -p "$OUT"
export SDT_HIP_LIB=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_tuning.so
SDT_SK_SKEW=120 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or stream or persist or sk" > "$OUT/pytest_skew120.txt" 2>&1
tail -n 3 "$OUT/pytest_skew120.txt"
for sk in 0 60 120 180 0 120; do
  echo "== skew $sk" >> "$OUT/conv_bench.txt"
  SDT_SK_SKEW=$sk python tools/conv_bench.py --tuning --roles fwd,dX --streamk 2 --only L2,L3,L4,L5,L6,L7 2>&1 | grep -v amdgpu.ids >> "$OUT/conv_bench.txt"
done
python - <<'PY'
import re
rows = {}; lab = None; order = []
for ln in open("gpurun_out/r04_skew/conv_bench.txt"):
    if ln.startswith("=="):
        lab = ln.split()[2] + ("b" if ln.split()[2] in order else ""); order.append(lab); continue
    m = re.match(r"(L\d)\s+(\w+)\s+([\d.]+) us", ln)
    if m: rows.setdefault((m.group(1), m.group(2)), {})[lab] = float(m.group(3))
print("layer role  " + "  ".join("%7s" % o for o in order))
tot = {o: 0.0 for o in order}
for k in sorted(rows):
    print("%s %-4s  " % k + "  ".join("%7.1f" % rows[k].get(o, float("nan")) for o in order))
    for o in order: tot[o] += rows[k].get(o, 0.0)
print("sum       " + "  ".join("%7.1f" % tot[o] for o in order))
PY
