#!/bin/bash
# round 5: the tuning-library tests against the final sources, the other configs on the final engine, the data-parallel A/B on one GPU
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_others
rm -rf "$OUT"; mkdir -p "$OUT"
SDT_HIP_LIB=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_tuning.so timeout 900 python -m pytest tests -q -m "gpu and tuning" > "$OUT/pytest_tuning.txt" 2>&1; tail -n 2 "$OUT/pytest_tuning.txt"
for cfg in voice2pose_sdt_vae voice2pose_s2g pose2pose; do
  for extra in "" "--graph"; do
    timeout 600 python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$extra', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" | tee -a "$OUT/bench.txt"
  done
done
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events --no-f32-split 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('voice2pose_sdt_bp --no-f32-split', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" | tee -a "$OUT/bench.txt"
bash tools/debug/r05_dp_ab.sh > "$OUT/dp.log" 2>&1; tail -16 "$OUT/dp.log"
