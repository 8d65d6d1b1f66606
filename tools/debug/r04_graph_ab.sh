cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for f in "" "--graph" "--graph --graph-streams"; do
  echo "== fp32 $f"
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode --no-kernel-events $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('streamk_errors'))"
done; done
