cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAYERS=${1:-L2,L3}
rm -rf gpurun_out/pmc_b1 gpurun_out/pmc_b2 gpurun_out/pmc_b3
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_b1 -o p -- python tools/bf16_conv_bench.py ${BENCH_ARGS:-} --layers $LAYERS --rep 3 --no-dw > gpurun_out/pmc_b1.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b1/p_counter_collection.csv --match conv --min-us 20 > gpurun_out/r5_pmc_bf2.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_b2 -o p -- python tools/bf16_conv_bench.py ${BENCH_ARGS:-} --layers $LAYERS --rep 3 --no-dw > gpurun_out/pmc_b2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b2/p_counter_collection.csv --match conv --min-us 20 >> gpurun_out/r5_pmc_bf2.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmc_b3 -o p -- python tools/bf16_conv_bench.py ${BENCH_ARGS:-} --layers $LAYERS --rep 3 --no-dw > gpurun_out/pmc_b3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b3/p_counter_collection.csv --match conv --min-us 20 >> gpurun_out/r5_pmc_bf2.txt
rm -rf gpurun_out/pmc_b1 gpurun_out/pmc_b2 gpurun_out/pmc_b3
cat gpurun_out/r5_pmc_bf2.txt
