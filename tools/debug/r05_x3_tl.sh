# split-fp32 kernel: timeline of L2 / L5 forward + dX, LDS bank conflicts and issue / stall split of the launches
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/debug/sk_timeline.py --dtype f32 --bf2 --only L2,L5 --roles fwd,dX 2>&1 | grep -E "^L|K loop|fill|end|set-up|start|finish"
rm -rf gpurun_out/pmc_x
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_x -o p -- python tools/bf16_conv_bench.py --dtype f32 --layers L2,L5 --rep 3 --no-dw > gpurun_out/pmc_x.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_x/p_counter_collection.csv --match convbf2 --min-us 20
rm -rf gpurun_out/pmc_x
