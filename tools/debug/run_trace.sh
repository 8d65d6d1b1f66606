cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02_trace7
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02_trace7 -o t -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/r02_trace7.log 2>&1
f=$(find gpurun_out/r02_trace7 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'c1d_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
print('c1d launches',n)
per=33
last=rows[-per:]
for r in last:
    print(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['LDS_Block_Size'] if 'LDS_Block_Size' in r else '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
rm -rf gpurun_out/r02_trace7/*/*.db 2>/dev/null
tail -3 gpurun_out/r02_trace7.log
