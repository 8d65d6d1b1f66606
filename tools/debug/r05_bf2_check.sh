# correctness + per-layer timing + timeline of the bf16-shaped kernel (convbf.hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
timeout 1200 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "conv_kernels or backward_statistics" > gpurun_out/r5_bf2_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_bf2_pytest.txt
tail -5 gpurun_out/r5_bf2_pytest.txt
fi
timeout 300 python tools/bf16_conv_bench.py --rep 10 > gpurun_out/r5_bf2_layers.txt 2>&1
cat gpurun_out/r5_bf2_layers.txt
timeout 600 python tools/debug/sk_timeline.py --dtype bf16 --bf2 --only ${TL:-L2,L4} --roles fwd,dX > gpurun_out/r5_bf2_timeline.txt 2>&1
cat gpurun_out/r5_bf2_timeline.txt
rm -f gpurun_out/sk_tl_*.npy
