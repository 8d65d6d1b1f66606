cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/bin/mfma_peak 2000 > gpurun_out/mfma_peak2.txt 2>&1
for p in 0 31 32 33 0 31; do
  echo "== SDT_CONV_PRIO=$p" >> gpurun_out/prio_ab.txt
  SDT_CONV_PRIO=$p timeout 600 python tools/conv_bench.py --tuning --only L1,L2,L3,L4,L5,L6,L7 --roles fwd,dX --reps 20 >> gpurun_out/prio_ab.txt 2>&1
done
SDT_CONV_PRIO=31 timeout 600 python tools/debug/taps_timeline.py --only L1,L2,L4 --roles fwd > gpurun_out/timeline_p31.txt 2>&1
SDT_CONV_PRIO=32 timeout 600 python tools/debug/taps_timeline.py --only L1,L2,L4 --roles fwd > gpurun_out/timeline_p32.txt 2>&1
