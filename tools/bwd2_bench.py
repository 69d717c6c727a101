import os, sys, torch
sys.path.insert(0, os.getcwd())
from miso_amd import ops
dev="cuda:0"
n=262144; L,C=3,8
torch.manual_seed(0)
feats=[(torch.randn(1,C,s,s,s,device=dev)*1e-2).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) for s in (32,64,128)]
meta=ops.GridMeta.from_bound([[-1.,1.]]*3)
W=torch.randn(L*C,1,device=dev)
x=(torch.rand(n,3,device=dev)*2-1)
def step():
    xd=x.clone().requires_grad_(True)
    f=ops.encode(xd,feats,meta)
    s=(f@W)
    g,=torch.autograd.grad(s.sum(),xd,create_graph=True)
    loss=((g.norm(dim=1)-1)**2).mean()
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print("eikonal step (encode fwd + bwd_x + bwd2) per iter: %.1f us" % (e0.elapsed_time(e1)/10*1e3))
