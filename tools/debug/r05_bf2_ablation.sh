# K-loop ablation of the bf16-shaped kernel: libraries built with -DBF_ABL=<bits> (wrong results by design), per-layer times each.
# BENCH_ARGS="--dtype f32": the split-fp32 form.  build (in the build container):  bash tools/debug/r05_bf2_ablation.sh build ;  run (GPU box): bash tools/debug/r05_bf2_ablation.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=speechdrivestemplates_amd/lib
ABLS="${ABLS:-0 1 2 4 8 16 3 19 23}"
if [ "$1" = "build" ]; then
  for a in $ABLS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DBF_ABL=$a -c speechdrivestemplates_amd/csrc/convbf.hip -o /tmp/convbf_abl$a.o &
  done
  wait
  for a in $ABLS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libsdt_hip_abl$a.so $L/conv.o $L/convsk.o /tmp/convbf_abl$a.o $L/norm.o $L/misc.o $L/l0.o $L/chain1d.o
  done
  ls -la $L/*.so
  exit 0
fi
mkdir -p gpurun_out
rm -f gpurun_out/r5_bf2_ablation.txt
for a in $ABLS; do
  echo "== BF_ABL=$a (1 no global loads, 2 no LDS stores, 4 no MFMA, 8 no barrier, 16 no fragment reads)" >> gpurun_out/r5_bf2_ablation.txt
  SDT_HIP_LIB=$PWD/$L/libsdt_hip_abl$a.so SDT_ALLOW_NAN=1 timeout 300 python tools/bf16_conv_bench.py --rep ${REP:-10} --no-dw ${BENCH_ARGS:-} ${LAYERS:+--layers $LAYERS} 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_bf2_ablation.txt
done
cat gpurun_out/r5_bf2_ablation.txt
