cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "single_items or reserved_slots or weight_grad or dw" 2>&1 | tail -8
for sp in 0 1; do echo "split $sp"; timeout 300 python tools/bf16_conv_bench.py --dtype f32 --split $sp --rep 5 2>&1 | grep -E "^L|^sum"; done
