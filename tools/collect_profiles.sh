#!/bin/bash
# Round evidence, run on the MI355X box: bash tools/collect_profiles.sh r03   (writes gpurun_out/<tag>_final/)
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_final
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-mode"
# (1) the bench line as the driver runs it
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# (2) kernel trace + stats of the same command (default: weight gradients on the side stream)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o b -- $CMD > "$OUT/trace.log" 2>&1
# (3) every kernel alone on the GPU
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_noovl" -o b -- $CMD --no-overlap-dw --no-kernel-events > "$OUT/trace_noovl.log" 2>&1
python tools/trace_summary.py "$OUT/trace_noovl/b_kernel_trace.csv" 35 60 > "$OUT/trace_by_launch_shape.txt" 2>&1
# trace-based fraction of the dominant kernel (bench.py cites it beside its event-based figure): algorithmic GFLOP per launch from the bench line
GF=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['roofline']['algorithmic_gflop_per_launch'])")
# (the dominant kernel and its matrix peak as the bench line names them; "name<a, b>" of the bench = "name<a, b, ..." of the trace)
KN=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
PK=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['roofline']['peak'])")
RX=$(python -c "import re,sys; k=sys.argv[1]; print(re.escape(k[:-1]) + '[,>]' if k.endswith('>') else re.escape(k))" "$KN")
case "$KN" in "convsk_kernel<"*) RX="convsk_kernel<float, float, ${KN#convsk_kernel<}"; RX="${RX%>}";; esac
python tools/trace_fraction.py "$OUT/trace_noovl/b_kernel_trace.csv" "$RX" "$KN" "$GF" "$PK" "$OUT/trace_fraction.json" > "$OUT/trace_fraction.log" 2>&1
python tools/stream_summary.py "$OUT/trace/b_kernel_trace.csv" 35 14 > "$OUT/streams.txt" 2>&1
python tools/hbm_kernels.py "$OUT/trace_noovl/b_kernel_trace.csv" 35 > "$OUT/hbm_kernels.txt" 2>&1
# (3a) the bf16-storage step (BASELINE config 4's arithmetic): every kernel alone, and as run
BCMD="python bench.py --storage bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-kernel-events"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bf16" -o b -- $BCMD --no-overlap-dw > "$OUT/trace_bf16.log" 2>&1
python tools/trace_summary.py "$OUT/trace_bf16/b_kernel_trace.csv" 25 60 > "$OUT/bf16_trace_by_launch_shape.txt" 2>&1
cp "$OUT/trace_bf16/b_kernel_stats.csv" "$OUT/bf16_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/trace_bf16"
# (3b) PMC of the MFMA kernels, one launch set per layer and role (two passes: issue / stall split, instruction mix)
bash tools/debug/pmc_conv.sh L1,L2,L3,L4,L5,L6,L7 fwd,dX,dW > "$OUT/pmc_conv.txt" 2>&1
# (4) fabric-side traffic: two PMC passes (FETCH_SIZE / WRITE_SIZE cannot share one)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o b -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-alt-mode > "$OUT/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o b -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-alt-mode > "$OUT/pmc_write.log" 2>&1
python tools/hbm_traffic.py "$OUT/pmc_fetch/b_counter_collection.csv" "$OUT/pmc_write/b_counter_collection.csv" "$OUT/hbm_traffic_bench.json" "python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-alt-mode" "$KN" > "$OUT/hbm_traffic.log" 2>&1
# keep the merged payload small: drop the raw traces / databases, keep the stats
for d in trace trace_noovl; do cp "$OUT/$d/b_kernel_stats.csv" "$OUT/${d}_kernel_stats.csv" 2>/dev/null; done
rm -rf "$OUT/trace" "$OUT/trace_noovl" "$OUT/pmc_fetch" "$OUT/pmc_write"
tail -c 300 "$OUT/bench.json"; echo; tail -3 "$OUT/hbm_traffic.log"; head -12 "$OUT/hbm_kernels.txt"
