#!/bin/bash
# ordered kernel sequence of one fp32 train step.   gpurun -- bash tools/debug/r04_seq.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04i
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04i/trace -o b -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-alt-mode --no-kernel-events > gpurun_out/r04i/trace.log 2>&1
python tools/debug/step_sequence.py gpurun_out/r04i/trace/b_kernel_trace.csv 3 > gpurun_out/r04i/sequence.txt 2>&1
rm -rf gpurun_out/r04i/trace
wc -l gpurun_out/r04i/sequence.txt
