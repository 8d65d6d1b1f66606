cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} > gpurun_out/r5_full_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_full_pytest.txt
tail -15 gpurun_out/r5_full_pytest.txt
if [ "$NOBENCH" = "" ]; then
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_full_bench.txt 2>&1
tail -1 gpurun_out/r5_full_bench.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('fp32', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms; roofline', round(d['roofline']['frac'],4))
a=d.get('alt_conv_math')
print('bf16', round(a['value'],1), 'clips/s', round(a['ms_per_step'],3), 'ms eager', round(a['eager_ms_per_step'],3), a.get('roofline',{}).get('all_conv2d_launches'))
print({k: round(v['ms_per_step'],3) for k,v in d['conv_kernels'].items()})
"
fi
