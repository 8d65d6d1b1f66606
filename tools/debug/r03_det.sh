cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "repeat_bit or trajectories or train_step" -x -s > gpurun_out/det_test.txt 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x > gpurun_out/ops_test.txt 2>&1
