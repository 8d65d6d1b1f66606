cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug/sk_timeline.py --only L2,L4 --roles fwd --wpc 1 > gpurun_out/sk_timeline1.txt 2>&1
timeout 600 python tools/debug/sk_timeline.py --only L2,L4 --roles fwd --wpc 2 > gpurun_out/sk_timeline2.txt 2>&1
rm -f gpurun_out/sk_bench.txt
for v in "--streamk 1" "--streamk 1 --oob" "--streamk 2"; do
echo "== $v" >> gpurun_out/sk_bench.txt
SDT_SK_ALL=1 timeout 600 python tools/conv_bench.py $v --only L1,L2,L3,L4,L5,L6,L7 --roles fwd,dX --reps 20 >> gpurun_out/sk_bench.txt 2>&1
done
