cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
python tests/tools/dp_graph_case.py --out /tmp/x.npz --port 2961$i --config voice2pose_s2g --storage f32 --mode split > gpurun_out/r5_s2g_split_$i.txt 2>&1
echo "rc $?" >> gpurun_out/r5_s2g_split_$i.txt
done
python tests/tools/dp_graph_case.py --out /tmp/x.npz --port 29619 --config voice2pose_s2g --storage f32 --mode full > gpurun_out/r5_s2g_full.txt 2>&1
echo "rc $?" >> gpurun_out/r5_s2g_full.txt
grep -n "rc \|what\|Error\|error" gpurun_out/r5_s2g_split_1.txt gpurun_out/r5_s2g_split_2.txt gpurun_out/r5_s2g_full.txt | head -20
