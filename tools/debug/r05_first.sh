cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_dp_gpu.py tests/test_chain1d_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "dp_gpu or chain or exchange or hipgraph or repeat_bit" > gpurun_out/r5_first_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_first_pytest.txt
SDT_HIP_LIB=$GRAFT_REPO_ROOT/speechdrivestemplates_amd/lib/libsdt_hip_tuning.so timeout 600 python -m pytest tests/test_chain1d_gpu.py tests/test_ops_gpu.py -m gpu -q -k "lost" > gpurun_out/r5_first_tuning.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_first_bench.txt 2>&1
SDT_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_first_bench_dp1.txt 2>&1
tail -3 gpurun_out/r5_first_pytest.txt; tail -3 gpurun_out/r5_first_tuning.txt
