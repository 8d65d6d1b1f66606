#!/bin/bash
# builds the ablation libraries of tools/debug/r04_bf16_ablation.sh (wrong results by design; never loaded by the package)
set -e
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
for a in 1 2 4 3 7; do
  /opt/rocm/bin/hipcc $F -DSK_BF_ABL=$a -c speechdrivestemplates_amd/csrc/convsk.hip -o /tmp/convsk_abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o speechdrivestemplates_amd/lib/libsdt_hip_abl$a.so /tmp/convsk_abl$a.o speechdrivestemplates_amd/lib/conv.o speechdrivestemplates_amd/lib/norm.o speechdrivestemplates_amd/lib/misc.o speechdrivestemplates_amd/lib/l0.o
done
ls -la speechdrivestemplates_amd/lib/
