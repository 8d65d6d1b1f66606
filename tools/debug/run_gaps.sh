cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02_gaps
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02_gaps -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > gpurun_out/r02_gaps.log 2>&1
python tools/queue_gaps.py gpurun_out/r02_gaps/t_kernel_trace.csv 25 6; python tools/debug/step_timeline.py gpurun_out/r02_gaps/t_kernel_trace.csv 40
rm -rf gpurun_out/r02_gaps
