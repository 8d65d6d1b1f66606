# split-fp32 form of the 8-wave conv kernel: accuracy against float64 / the fp32-MFMA kernels, per-layer time, timeline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/debug/x3_check.py --batch ${XB:-8} 2>&1 | tail -9
for sp in ${SPLITS:-1}; do
echo "split $sp"
timeout 300 python tools/bf16_conv_bench.py --dtype f32 --split $sp --rep 5 --no-dw 2>&1 | grep -E "^L|^sum"
done
timeout 300 python tools/debug/sk_timeline.py --dtype f32 --bf2 --only ${TL:-L2,L5} --roles fwd,dX 2>&1 | grep -E "^L|K loop per|fill|end  |set-up"
