# re-record tests/golden/margins.json + the parity tables on the MI355X box (outputs under gpurun_out/; copy margins.json to tests/golden/ afterwards)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.json gpurun_out/parity_tables.txt
SDT_PARITY_TABLES=gpurun_out/parity_tables.txt SDT_RECORD_MARGINS=gpurun_out/margins.json SDT_RECORD_MARGINS_MAX=1 timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_rec1.txt 2>&1
SDT_RECORD_MARGINS=gpurun_out/margins.json SDT_RECORD_MARGINS_MAX=1 timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q > gpurun_out/pytest_rec2.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1
