#!/bin/bash
# Conv1d chain launch A/B on the headline config.   gpurun -- bash tools/debug/r04_chain_ab.sh
mkdir -p gpurun_out/r04h
for rep in 1 2; do
for f in "" "--no-chain1d"; do
  echo "== fp32 $f" >> gpurun_out/r04h/chain_ab.txt
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('streamk_errors'))" >> gpurun_out/r04h/chain_ab.txt
done; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r04h/bench_driver.txt 2> gpurun_out/r04h/bench_driver.err
cat gpurun_out/r04h/chain_ab.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04h/bench_driver.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_conv1d"]["windows_us"], d["alt_conv_math"]["value"], d["alt_conv_math"]["ms_per_step"])
PY
