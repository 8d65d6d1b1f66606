# round-3 diagnostics, one gpurun call: MFMA pipe ceiling, per-workgroup timelines of conv_taps, per-stage forward errors, PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/bin/mfma_peak 2000 > gpurun_out/mfma_peak.txt 2>&1
timeout 600 python tools/debug/taps_timeline.py --only L1,L2,L4,L5,L6,L7 --roles fwd,dX > gpurun_out/timeline.txt 2>&1
rm -f gpurun_out/stage_errors.txt
SDT_PARITY_TABLES=gpurun_out/stage_errors.txt timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k stage_error_table > gpurun_out/stage_errors.log 2>&1
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err
bash tools/debug/pmc_conv.sh L1,L2,L3,L4,L5,L6,L7 fwd,dX,dW > gpurun_out/pmc_conv.txt 2>&1
