cd $GRAFT_REPO_ROOT
for v in "--no-streamk" "" "--no-streamk-dw" "--no-streamk" ""; do
echo "== bench $v" >> gpurun_out/bench_ab.txt
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-mode $v 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('value %.1f uninstr %.1f  ms %.3f  roofline %s  conv_total %s' % (j['value'], j['value_uninstrumented'], j['ms_per_step'], {k: (round(v,1) if isinstance(v,float) else v) for k,v in j['roofline'].items() if k in ('kernel','achieved','frac')}, j['conv_total']))
        print('   kernels', {k: (round(v['ms_per_step'],3), round(v['tflops'],1)) for k,v in j['conv_kernels'].items()})
    elif 'Error' in l or 'error' in l:
        print(l.rstrip())
" >> gpurun_out/bench_ab.txt 2>&1
done
