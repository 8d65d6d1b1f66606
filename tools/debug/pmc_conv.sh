cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_c1 -o p -- python tools/conv_bench.py --only ${1:-L1,L2} --roles ${2:-fwd,dX,dW} --reps 3 > gpurun_out/pmc_c1.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_c1/p_counter_collection.csv --match conv --min-us 50
rm -rf gpurun_out/pmc_c2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/pmc_c2 -o p -- python tools/conv_bench.py --only ${1:-L1,L2} --roles ${2:-fwd,dX,dW} --reps 3 > gpurun_out/pmc_c2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_c2/p_counter_collection.csv --match conv --min-us 50
tail -2 gpurun_out/pmc_c2.log
rm -rf gpurun_out/pmc_c1 gpurun_out/pmc_c2
