cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.txt 2>&1
timeout 600 python bench.py --steps 40 --warmup 10 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1
