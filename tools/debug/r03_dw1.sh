cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug/sk_check.py 32 sk > gpurun_out/sk_check.txt 2>&1
rm -f gpurun_out/sk_bench.txt
for v in "--streamk 0" "--streamk 2" "--streamk 1" "--streamk 0" "--streamk 2"; do
echo "== $v" >> gpurun_out/sk_bench.txt
timeout 600 python tools/conv_bench.py $v --only L1,L2,L3,L4,L5,L6,L7 --roles dW --reps 20 >> gpurun_out/sk_bench.txt 2>&1
done
