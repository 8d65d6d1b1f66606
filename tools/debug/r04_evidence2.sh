cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_ev2
python tools/debug/chain_timeline.py > gpurun_out/r04_ev2/chain_timeline.txt 2>&1
python tools/debug/chain_timeline.py --math bf16 > gpurun_out/r04_ev2/chain_timeline_bf16.txt 2>&1
python tools/host_time.py > gpurun_out/r04_ev2/host_time_f32.txt 2>&1
python tools/host_time.py --storage bf16 > gpurun_out/r04_ev2/host_time_bf16.txt 2>&1
python tools/host_time.py --config voice2pose_s2g > gpurun_out/r04_ev2/host_time_s2g.txt 2>&1
python tools/debug/comm_emulation.py --reserve 0 --us 600 --steps 25 > gpurun_out/r04_ev2/comm_emulation.txt 2>&1
python tools/debug/comm_emulation.py --reserve 32 --us 600 --steps 25 >> gpurun_out/r04_ev2/comm_emulation.txt 2>&1
for c in voice2pose_sdt_vae voice2pose_s2g "voice2pose_s2g --graph" pose2pose "pose2pose --graph"; do echo "== $c" >> gpurun_out/r04_ev2/other_configs.txt; python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/r04_ev2/other_configs.txt; done
tail -n 1 gpurun_out/r04_ev2/host_time_*.txt; cat gpurun_out/r04_ev2/other_configs.txt; grep -E "span|sum" gpurun_out/r04_ev2/chain_timeline*.txt; grep reserve gpurun_out/r04_ev2/comm_emulation.txt
