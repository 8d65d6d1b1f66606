cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAYERS=${1:-L2,L3}
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|TA|TD)_[A-Z0-9_a-z]+" | sort -u > gpurun_out/r5_counters_mem.txt
run() { tag=$1; shift; rm -rf gpurun_out/pmc_m; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_m -o p -- python tools/bf16_conv_bench.py --layers $LAYERS --rep 3 --no-dw > gpurun_out/pmc_m.log 2>&1; echo "## $tag" >> gpurun_out/r5_pmc_mem.txt; python tools/pmc_summary.py gpurun_out/pmc_m/p_counter_collection.csv --match conv --min-us 20 >> gpurun_out/r5_pmc_mem.txt 2>&1 || tail -3 gpurun_out/pmc_m.log >> gpurun_out/r5_pmc_mem.txt; }
rm -f gpurun_out/r5_pmc_mem.txt
run l2hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run l2ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_RD_UNCACHED_32B_sum
run tcplat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
run tcpstall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
rm -rf gpurun_out/pmc_m
cat gpurun_out/r5_pmc_mem.txt
