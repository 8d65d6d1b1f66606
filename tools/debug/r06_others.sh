cd $GRAFT_REPO_ROOT
for cfg in voice2pose_sdt_bp voice2pose_sdt_vae voice2pose_s2g pose2pose; do
  for extra in "" "--graph"; do
    timeout 600 python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$extra', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))"
  done
done
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events --no-f32-split 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('voice2pose_sdt_bp --no-f32-split', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))"
timeout 600 python bench.py --storage bf16 --graph --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('voice2pose_sdt_bp --storage bf16 --graph', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))"
