cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_chain1d_gpu.py -m gpu -x -q -s -k "float64_oracle" > gpurun_out/r5_t2_chain.txt 2>&1
echo "rc $?" >> gpurun_out/r5_t2_chain.txt
grep -E "chain vs float64|passed|failed|Error|assert" gpurun_out/r5_t2_chain.txt | head
for i in 1 2 3; do
python tests/tools/dp_graph_case.py --out /tmp/x.npz --port 2962$i --config voice2pose_s2g --storage f32 --mode split > gpurun_out/r5_s2g_split_$i.txt 2>&1
echo "split $i rc $?"
python tests/tools/dp_graph_case.py --out /tmp/x.npz --port 2963$i --config voice2pose_sdt_bp --storage bf16 --mode full > gpurun_out/r5_bp_full_$i.txt 2>&1
echo "full $i rc $?"
done
SDT_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_t2_bench_dp1.txt 2>&1
tail -1 gpurun_out/r5_t2_bench_dp1.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('dp1 eager', round(d['value'],1), 'graph leg', d.get('dp_graph_replay'), 'bf16', {k:d['alt_conv_math'][k] for k in ('value','ms_per_step','graph_mode')})
"
