mkdir -p gpurun_out/r04f
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-alt-mode --no-kernel-events"
for rep in 1 2; do
for v in "" "--bwd-wpc 1,1" "--bwd-wpc 2,1" "--bwd-wpc 1,2"; do
  echo "== fp32 $v" >> gpurun_out/r04f/ab.txt
  $B $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['median_ms_per_step'])" >> gpurun_out/r04f/ab.txt 2>&1
done
done
for v in "--graph" "--graph --graph-streams"; do
  echo "== bf16 $v" >> gpurun_out/r04f/ab.txt
  $B --storage bf16 $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['median_ms_per_step'])" >> gpurun_out/r04f/ab.txt 2>&1
done
cat gpurun_out/r04f/ab.txt
