#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/s1d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s1d -o p -- python tools/debug/small1d_check.py > gpurun_out/s1d.log 2>&1
python - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/s1d/p_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if n.startswith(("conv1d_small", "conv_taps_kernel", "splitk_reduce")):
        d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: (kv[0][1:], kv[0][0])):
    v.sort()
    print("%-45s grid %6s %2s %2s  n %4d  median %6.2f us  p10 %6.2f" % (k[0][:45], int(k[1]) // 256, k[2], k[3], len(v), v[len(v) // 2], v[len(v) // 10]))
PY
rm -rf gpurun_out/s1d
