# Evidence of the split-fp32 kernels (profiles/r05_split_f32_*.txt, r05_mfma_valu_probe.txt): accuracy table, per-layer times, timeline, MFMA / VALU probe
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_split; rm -rf $O; mkdir -p $O
( echo "# tools/debug/x3_check.py --batch 8: error against float64 of the fp32-MFMA kernels and of the split-fp32 kernel, same tensors"; timeout 600 python tools/debug/x3_check.py --batch 8 2>&1 | grep -E "^L|worst" ) > $O/accuracy.txt
( for sp in 0 1; do echo "# tools/bf16_conv_bench.py --dtype f32 --split $sp   (B = 32, us per launch and TFLOP/s of algorithmic fp32 FLOPs: forward | input gradient | weight gradient)"; timeout 300 python tools/bf16_conv_bench.py --dtype f32 --split $sp --rep 5 2>&1 | grep -E "^L|^sum"; done ) > $O/per_layer.txt
( echo "# tools/debug/sk_timeline.py --dtype f32 --bf2 (tuning build): per-segment stamps of the split-fp32 conv kernel"; timeout 300 python tools/debug/sk_timeline.py --dtype f32 --bf2 --only L2,L5 --roles fwd,dX 2>&1 | grep -vE "amdgpu.ids" ) > $O/timeline.txt
( echo "# tools/debug/mfma_valu_probe.hip (hipcc --offload-arch=gfx950 -O3): do VALU instructions hide under MFMAs on one SIMD?"; ./tools/debug/mfma_valu_probe.bin ) > $O/mfma_valu_probe.txt
cat $O/accuracy.txt $O/per_layer.txt $O/mfma_valu_probe.txt | cut -c1-220
