cd $GRAFT_REPO_ROOT
rm -f gpurun_out/sk_bench.txt
for k in 1 2; do
echo "== streamk $k real" >> gpurun_out/sk_bench.txt
timeout 600 python tools/conv_bench.py --streamk $k --only L1,L2,L4,L5,L6 --roles fwd,dX --reps 20 >> gpurun_out/sk_bench.txt 2>&1
echo "== streamk $k oob" >> gpurun_out/sk_bench.txt
timeout 600 python tools/conv_bench.py --streamk $k --oob --only L1,L2,L4,L5,L6 --roles fwd,dX --reps 20 >> gpurun_out/sk_bench.txt 2>&1
done
