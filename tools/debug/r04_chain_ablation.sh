#!/bin/bash
# K-loop ablation of the Conv1d chain kernels.  Build here (no GPU): bash tools/debug/r04_chain_ablation.sh build ; run on the box: ... run
set -e
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DSDT_TUNING"
L=speechdrivestemplates_amd/lib
if [ "$1" = build ]; then
  for src in conv convsk norm misc l0; do /opt/rocm/bin/hipcc $F -c speechdrivestemplates_amd/csrc/$src.hip -o /tmp/t_$src.o & done; wait
  for a in 0 1 2 4 6 7; do
    /opt/rocm/bin/hipcc $F -DCH_ABL=$a -c speechdrivestemplates_amd/csrc/chain1d.hip -o /tmp/chain_abl$a.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libsdt_hip_chabl$a.so /tmp/chain_abl$a.o /tmp/t_conv.o /tmp/t_convsk.o /tmp/t_norm.o /tmp/t_misc.o /tmp/t_l0.o
  done
  ls $L
else
  mkdir -p gpurun_out/r04j
  for a in 0 1 2 4 6 7; do
    echo "== CH_ABL=$a (1 no MFMA, 2 no weight loads in the loop, 4 no LDS stores in the loop)" >> gpurun_out/r04j/chain_ablation.txt
    SDT_CHAIN_LIB=$L/libsdt_hip_chabl$a.so python tools/debug/chain_timeline.py 2>&1 | grep -E "launch span|sum|dec3|e6 " >> gpurun_out/r04j/chain_ablation.txt
  done
  cat gpurun_out/r04j/chain_ablation.txt
fi
