# correctness + per-layer timing of the bf16-shaped kernel (convbf.hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
timeout 1200 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "conv_kernels or backward_statistics" > gpurun_out/r5_bf2_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_bf2_pytest.txt
tail -5 gpurun_out/r5_bf2_pytest.txt
fi
timeout 300 python tools/bf16_conv_bench.py --rep 10 --old > gpurun_out/r5_bf2_layers_old.txt 2>&1
cat gpurun_out/r5_bf2_layers_old.txt
timeout 300 python tools/bf16_conv_bench.py --rep 10 > gpurun_out/r5_bf2_layers.txt 2>&1
cat gpurun_out/r5_bf2_layers.txt
