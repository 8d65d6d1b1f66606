"""Debug: does the pose2pose B=4 step-0 gradient gap come from discrete events (L1 sign / LeakyReLU flips)?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from oracle import sdt_oracle as O
from test_model_gpu import _make_pipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pipe, _ = _make_pipeline("pose2pose", 16 if B == 4 else 64, 0.0)
ocfg = O.cfg_named("pose2pose")
batch = O.make_batch(B, 16 if B == 4 else 64, step=0, seed=1)
eps = torch.from_numpy(np.random.Generator(np.random.PCG64([2, 0])).standard_normal((B, 32)).astype(np.float32))
rr = torch.randn
torch.randn = lambda *a, **k: eps.clone().cuda()
losses, res = pipe.forward_backward(batch)
torch.randn = rr
torch.cuda.synchronize()
pred_h = res["poses_pred_batch"].detach().cpu()
gt = batch["poses"]
outs = {}
for dt in (torch.float32, torch.float64):
    st = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in O.make_pose2pose_state(ocfg, 16 if B == 4 else 64, seed=0).items()}
    b = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    O.OraclePose2Pose(ocfg, st)
    l, r = O.pose2pose_forward(st, b, ocfg, eps.to(dt), True)
    l["loss"].backward()
    outs[dt] = (r["poses_pred_batch"].detach(), {k: v.grad for k, v in st.items() if v.requires_grad and v.grad is not None})
p32, p64 = outs[torch.float32][0], outs[torch.float64][0]
def sgn(p): return torch.sign(p.double() - gt.double())
print("L1 sign mismatches  hip vs f64: %d   ref32 vs f64: %d   hip vs ref32: %d   (of %d)" % (
    (sgn(pred_h) != sgn(p64)).sum(), (sgn(p32) != sgn(p64)).sum(), (sgn(pred_h) != sgn(p32)).sum(), gt.numel()))
print("exact zeros of pred-gt: hip %d ref32 %d" % ((pred_h == gt).sum(), (p32 == gt).sum()))
d = (pred_h.double() - p64).abs()
print("pred err hip-f64 max %.3e  ref32-f64 max %.3e" % (d.max(), (p32.double() - p64).abs().max()))
gh = {k: p.grad.detach().double().cpu() for k, p in pipe.model.named_parameters() if p.grad is not None}
for k in ("ae.decoder.blocks.4.bias", "ae.decoder.blocks.4.weight", "ae.decoder.blocks.3.norm.bias", "ae.decoder.blocks.3.conv.weight", "ae.encoder.blocks.0.conv.weight"):
    g64 = outs[torch.float64][1][k]; g32 = outs[torch.float32][1][k].double()
    print("%-34s hip %.3e ref32 %.3e  (max|g| %.3e, 1/N %.3e)" % (k, (gh[k] - g64).abs().max() / g64.abs().max(), (g32 - g64).abs().max() / g64.abs().max(), g64.abs().max(), 1.0 / gt.numel()))
k = "ae.decoder.blocks.4.bias"
dd = (gh[k] - outs[torch.float64][1][k]).abs()
print("bias grad abs err top5 (in units of 1/N):", (dd.topk(5).values * gt.numel()).tolist())
