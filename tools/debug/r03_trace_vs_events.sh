#!/bin/bash
# does rocprofv3 --kernel-trace slow the persistent kernels?  same process: HIP-event average (conv_bench) vs traced kernel durations
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== plain"
timeout 300 python tools/conv_bench.py --only L4,L5,L6 --roles fwd,dW 2>&1 | grep -v amdgpu
echo "== under rocprofv3 --kernel-trace"
rm -rf gpurun_out/tve
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tve -o p -- python tools/conv_bench.py --only L4,L5,L6 --roles fwd,dW 2>&1 | grep -E "^L[0-9]"
python - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/tve/p_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if n.startswith("convsk"):
        d[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in d.items():
    v.sort()
    durs = [(e - s) / 1e3 for s, e in v]
    gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    print("%-40s n %3d  traced duration: min %.1f  median %.1f  max %.1f us;  gap to the next launch of this kernel: median %.1f us" % (n, len(v), min(durs), sorted(durs)[len(durs) // 2], max(durs), sorted(gaps)[len(gaps) // 2]))
PY
rm -rf gpurun_out/tve
