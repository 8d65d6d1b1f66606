#!/bin/bash
# the tuning-library tests against the final sources, then the other configs on the final engine
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r04_others
mkdir -p "$OUT"
SDT_HIP_LIB=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_tuning.so timeout 900 python -m pytest tests -q -m "gpu and tuning" > "$OUT/pytest_tuning.txt" 2>&1; tail -n 2 "$OUT/pytest_tuning.txt"
for cfg in voice2pose_sdt_vae voice2pose_s2g pose2pose; do
  for extra in "" "--graph"; do
    python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$extra', round(d['value'],1), round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a "$OUT/bench.txt"
  done
done
