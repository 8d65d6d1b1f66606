cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_pre3
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_pre3 -o p -- python tools/pre_bench.py --only L3,L4 --reps 2 > gpurun_out/pmc_pre3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_pre3/p_counter_collection.csv --match conv_taps --min-us 50
rm -rf gpurun_out/pmc_pre4
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/pmc_pre4 -o p -- python tools/pre_bench.py --only L3,L4 --reps 2 > gpurun_out/pmc_pre4.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_pre4/p_counter_collection.csv --match conv_taps --min-us 50
tail -3 gpurun_out/pmc_pre4.log
rm -rf gpurun_out/pmc_pre3/*.db gpurun_out/pmc_pre4/*.db
