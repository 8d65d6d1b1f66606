#!/bin/bash
# round-4 evidence that tools/collect_profiles.sh does not cover.   gpurun -- bash tools/debug/r04_evidence.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_ev
./tools/bin/xcd_barrier_probe > gpurun_out/r04_ev/xcd_barrier_probe.txt 2>&1
./tools/bin/tr16_probe > gpurun_out/r04_ev/tr16_probe.txt 2>&1
python tools/debug/chain_timeline.py > gpurun_out/r04_ev/chain_timeline.txt 2>&1
rm -f gpurun_out/r04j/chain_ablation.txt
bash tools/debug/r04_chain_ablation.sh run > /dev/null 2>&1
cp gpurun_out/r04j/chain_ablation.txt gpurun_out/r04_ev/chain_ablation.txt
python tools/debug/comm_emulation.py --reserve 0 --us 600 --steps 25 > gpurun_out/r04_ev/comm_emulation.txt 2>&1
python tools/debug/comm_emulation.py --reserve 32 --us 600 --steps 25 >> gpurun_out/r04_ev/comm_emulation.txt 2>&1
python tools/debug/comm_emulation.py --reserve 32 --us 1200 --steps 25 >> gpurun_out/r04_ev/comm_emulation.txt 2>&1
python tools/host_time.py > gpurun_out/r04_ev/host_time_f32.txt 2>&1
python tools/host_time.py --storage bf16 > gpurun_out/r04_ev/host_time_bf16.txt 2>&1
bash tools/debug/record_margins.sh
tail -n 3 gpurun_out/pytest_rec1.txt gpurun_out/pytest_rec2.txt gpurun_out/smoke.txt
