# PMC view of the bf16-storage conv kernels (per layer, each launch alone): MFMA busy, wait split, instruction mix, LDS activity
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAYERS=${1:-L2,L4,L6}
timeout 300 python tools/bf16_conv_bench.py --rep 20 > gpurun_out/r5_bf16_layers.txt 2>&1
rm -rf gpurun_out/pmc_b1 gpurun_out/pmc_b2 gpurun_out/pmc_b3
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_b1 -o p -- python tools/bf16_conv_bench.py --layers $LAYERS --rep 3 > gpurun_out/pmc_b1.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b1/p_counter_collection.csv --match conv --min-us 20 > gpurun_out/r5_pmc_bf16.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_b2 -o p -- python tools/bf16_conv_bench.py --layers $LAYERS --rep 3 > gpurun_out/pmc_b2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b2/p_counter_collection.csv --match conv --min-us 20 >> gpurun_out/r5_pmc_bf16.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmc_b3 -o p -- python tools/bf16_conv_bench.py --layers $LAYERS --rep 3 > gpurun_out/pmc_b3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b3/p_counter_collection.csv --match conv --min-us 20 >> gpurun_out/r5_pmc_bf16.txt
tail -3 gpurun_out/pmc_b3.log >> gpurun_out/r5_pmc_bf16.txt
rm -rf gpurun_out/pmc_b1 gpurun_out/pmc_b2 gpurun_out/pmc_b3
# the plain bench under the launcher (no process group): does the launcher's environment alone cost anything?
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain_under_launcher', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms')" > gpurun_out/r5_launcher.txt
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms')" >> gpurun_out/r5_launcher.txt
cat gpurun_out/r5_bf16_layers.txt; cat gpurun_out/r5_launcher.txt
