#!/bin/bash
# What bounds the K loop of the bf16 stream-K kernel?  Three builds that each REMOVE one instruction class from the loop (wrong results by design),
# timed by a kernel trace of the bf16 forward launches.  Libraries are built in the authoring container:
#   for a in 1 2 4 7; do hipcc ... -DSK_BF_ABL=$a ... -o speechdrivestemplates_amd/lib/libsdt_hip_abl$a.so; done   (tools/debug/r04_build_ablation.sh)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_abl; mkdir -p $OUT
for a in ${ABLS:-0 1 2 4 3 7}; do
  LIB=$PWD/speechdrivestemplates_amd/lib/libsdt_hip_abl$a.so
  [ $a = 0 ] && LIB=$PWD/speechdrivestemplates_amd/lib/libsdt_hip.so
  SDT_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$a -o b -- python tools/bf16_conv_bench.py --rep 10 > $OUT/log$a.txt 2>&1
  echo "== ablation $a (1: no global loads, 2: no LDS stores, 4: no MFMA)"
  python tools/trace_summary.py $OUT/t$a/b_kernel_trace.csv 1 40 2>/dev/null | grep -E "convsk_kernelIDF16b|convbf_dw" | head -12
  rm -rf $OUT/t$a
done
