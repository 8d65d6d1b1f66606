cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/debug/sk_timeline.py --dtype bf16 --bf2 --only ${1:-L2,L4,L6} --roles fwd,dX > gpurun_out/r5_bf2_timeline.txt 2>&1
cat gpurun_out/r5_bf2_timeline.txt
rm -f gpurun_out/sk_tl_*.npy
