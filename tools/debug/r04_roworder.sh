#!/bin/bash
# A/B of the stream-K kernel's K-step order (kernel row -> channel chunk -> tap, against tap -> channel chunk): parity tests first, then
# per-layer rates and the fabric traffic of the forward / input-gradient launches in both orders (tuning build: SDT_SK_ROW_ORDER=0 is the old order)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_roworder
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > "$OUT/pytest_ops.txt" 2>&1
tail -n 3 "$OUT/pytest_ops.txt"
for ro in 1 0 1 0; do
  echo "== SDT_SK_ROW_ORDER=$ro" >> "$OUT/conv_bench.txt"
  SDT_SK_ROW_ORDER=$ro timeout 600 python tools/conv_bench.py --tuning --roles fwd,dX --streamk 2 --only L2,L3,L4,L5,L6,L7 >> "$OUT/conv_bench.txt" 2>&1
done
for ro in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    SDT_SK_ROW_ORDER=$ro timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${ro}_$c" -o b -- python tools/conv_bench.py --tuning --roles fwd,dX --streamk 2 --only L2,L3,L4,L5,L6,L7 --reps 3 > "$OUT/pmc_${ro}_$c.log" 2>&1
  done
  python tools/hbm_traffic.py "$OUT/pmc_${ro}_FETCH_SIZE/b_counter_collection.csv" "$OUT/pmc_${ro}_WRITE_SIZE/b_counter_collection.csv" "$OUT/traffic_$ro.json" "conv_bench ro=$ro" > "$OUT/traffic_$ro.log" 2>&1
  rm -rf "$OUT/pmc_${ro}_FETCH_SIZE" "$OUT/pmc_${ro}_WRITE_SIZE"
done
cat "$OUT/conv_bench.txt" | tail -n 80
