#!/bin/bash
# FETCH_SIZE / duration of the stream-K forward launches with the n-tile-major tile order on / off (tuning library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for nb in 2097152 2000000000; do
  echo "== ntmajor threshold $nb bytes"
  rm -rf gpurun_out/pmc_nt
  SDT_SK_NTMAJOR=$nb timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_nt -o p -- python tools/conv_bench.py --tuning --only L4,L5,L6,L7 --roles fwd,dX --reps 3 > gpurun_out/pmc_nt.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_nt/p_counter_collection.csv --match convsk --min-us 50 --per-dispatch 2>&1 | head -40
  SDT_SK_NTMAJOR=$nb timeout 300 python tools/conv_bench.py --tuning --only L4,L5,L6,L7 --roles fwd,dX 2>&1 | grep -v amdgpu
done
rm -rf gpurun_out/pmc_nt
