import sys, torch
sys.path.insert(0, '.')
from __graft_entry__ import make_pipeline
from oracle import sdt_oracle as O
n_clips, B = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2
ocfg = O.cfg_named("voice2pose_sdt_bp")
state = O.make_voice2pose_state(ocfg, n_clips, seed=0, code_std=0.5)
pipe, _ = make_pipeline("voice2pose_sdt_bp", n_clips, state={k: v.clone() for k, v in state.items()})
eng = O.OracleVoice2Pose(ocfg, state)
batch = O.make_batch(B, n_clips, step=0, seed=1)
losses, results = pipe.forward_backward(batch)
grads = {k: p.grad.detach().cpu().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
pipe.optimizer_updates(losses)
# oracle grads
l2, r2 = O.voice2pose_forward(state, batch, ocfg, True)
for v in state.values():
    if v.requires_grad: v.grad = None
l2["G_loss"].backward()
rows = []
for k, g in grads.items():
    ref = state[k].grad
    if ref is None: continue
    d = (g - ref).abs().max().item(); s = ref.abs().max().item()
    # count sign disagreements among elements with |ref| > 1e-3*max
    big = ref.abs() > 1e-3 * s
    flips = ((torch.sign(g) != torch.sign(ref)) & big).sum().item()
    rows.append((d / max(s, 1e-30), k, s, flips, int(big.sum())))
rows.sort(reverse=True)
for r in rows[:12]:
    print("%-55s rel-err %.3e  max|g| %.3e  sign flips among significant %d/%d" % (r[1], r[0], r[2], r[3], r[4]))
