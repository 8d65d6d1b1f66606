# staging depth 3 / 4 (requests 3-4 K steps ahead of their LDS store): correctness, per-layer time with the tile interleave on and off
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "conv_kernels or backward_statistics" 2>&1 | tail -2
for il in 1 0; do
echo "interleave $il"
timeout 300 python tools/bf16_conv_bench.py --rep 10 --interleave $il 2>&1 | grep -E "^L|^sum"
done
timeout 300 python tools/debug/sk_timeline.py --dtype bf16 --bf2 --only L2,L5 --roles fwd 2>&1 | grep -E "^L|K loop|fill|end"
