#!/bin/bash
# PMC of the first block's kernels (tools/l0_time.py): instruction counts and busy / wait cycles of l0_bwd_sums_kernel and friends
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_pmc_l0
rm -rf gpurun_out/pmc_l1 gpurun_out/pmc_l2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_l1 -o p -- python tools/l0_time.py > gpurun_out/r04_pmc_l0/run1.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_l1/p_counter_collection.csv --match l0_ --min-us 5 > gpurun_out/r04_pmc_l0/pmc_l0.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_l2 -o p -- python tools/l0_time.py > gpurun_out/r04_pmc_l0/run2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_l2/p_counter_collection.csv --match l0_ --min-us 5 >> gpurun_out/r04_pmc_l0/pmc_l0.txt 2>&1
rm -rf gpurun_out/pmc_l1 gpurun_out/pmc_l2
cat gpurun_out/r04_pmc_l0/pmc_l0.txt
