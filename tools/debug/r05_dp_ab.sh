# data-parallel overhead on ONE GPU (1-rank RCCL group, SDT_DP_FORCE): what the reducer machinery costs before any byte crosses a link
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_dp_gpu.py -m gpu -x -q -k "two_ranks_one_gpu_bf16 or lost_partner" > gpurun_out/r5_dp_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_dp_pytest.txt
run() { tag=$1; shift; SDT_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" >> gpurun_out/r5_dp_ab.txt; }
rm -f gpurun_out/r5_dp_ab.txt
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" >> gpurun_out/r5_dp_ab.txt
run dp_default
run dp_reserve0 --dp-reserve 0
run dp_reserve16 --dp-reserve 16
run dp_nooverlap --dp-no-overlap
run dp_nooverlap_reserve0 --dp-no-overlap --dp-reserve 0
run dp_graph --graph
run dp_bf16_graph --storage bf16 --graph
run dp_bf16_eager --storage bf16
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode --storage bf16 --graph 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain_bf16_graph', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" >> gpurun_out/r5_dp_ab.txt
run dp_p2p_graph --config pose2pose --graph
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-alt-mode --config pose2pose --graph 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain_p2p_graph', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms  median', round(d['median_ms_per_step'],3))" >> gpurun_out/r5_dp_ab.txt
cat gpurun_out/r5_dp_ab.txt; tail -3 gpurun_out/r5_dp_pytest.txt
