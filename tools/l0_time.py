import sys, torch
sys.path.insert(0, ".")
from speechdrivestemplates_amd import ops
B,H,W=32,80,427
mel=torch.rand(B,H,W,device="cuda")
w=torch.nn.Parameter(ops.to_weight_layout(torch.randn(64,1,3,3,device="cuda")*0.3))
z=ops.L0BlockFn.apply(mel,w,None,None,None,None,None,B,0.2)
gz=torch.randn_like(z)
for name,fn in (("fwd",lambda: ops.L0BlockFn.apply(mel,w,None,None,None,None,None,B,0.2)),):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); print(name, e0.elapsed_time(e1)/20*1e3,"us")
outs=[ops.L0BlockFn.apply(mel,w,None,None,None,None,None,B,0.2) for _ in range(21)]
outs[0].backward(gz); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for o in outs[1:]: o.backward(gz)
e1.record(); torch.cuda.synchronize(); print("bwd", e0.elapsed_time(e1)/20*1e3,"us")
