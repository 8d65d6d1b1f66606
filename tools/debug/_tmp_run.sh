cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "trajectory or reserved_slots or split_f32" > gpurun_out/two.txt 2>&1; tail -5 gpurun_out/two.txt
BENCH_ARGS="--dtype f32" bash tools/debug/r05_pmc_bf2.sh L2,L5 2>&1 | tail -40
