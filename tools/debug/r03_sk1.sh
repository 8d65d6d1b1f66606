cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/debug/sk_check.py > gpurun_out/sk_check.txt 2>&1
for k in 0 1 2 0 1 2; do
echo "== streamk $k" >> gpurun_out/sk_bench.txt
timeout 600 python tools/conv_bench.py --streamk $k --only L1,L2,L3,L4,L5,L6,L7 --roles fwd,dX --reps 20 >> gpurun_out/sk_bench.txt 2>&1
done
