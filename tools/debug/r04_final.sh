#!/bin/bash
# final evidence of the round: the bench set, then the full GPU suite with margins / parity tables / smoke.   gpurun -- bash tools/debug/r04_final.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_ev
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
python tools/debug/chain_timeline.py > gpurun_out/r04_ev/chain_timeline.txt 2>&1
python tools/debug/chain_timeline.py --math bf16 > gpurun_out/r04_ev/chain_timeline_bf16.txt 2>&1
python tools/host_time.py > gpurun_out/r04_ev/host_time_f32.txt 2>&1
python tools/host_time.py --storage bf16 > gpurun_out/r04_ev/host_time_bf16.txt 2>&1
python tools/host_time.py --config voice2pose_s2g > gpurun_out/r04_ev/host_time_s2g.txt 2>&1
timeout 600 python tools/soak.py --steps 1500 > gpurun_out/r04_ev/soak.txt 2>&1
bash tools/debug/record_margins.sh
tail -n 2 gpurun_out/pytest_rec1.txt gpurun_out/pytest_rec2.txt gpurun_out/smoke.txt
