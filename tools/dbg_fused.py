import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from miso_amd import ops
dev="cuda:0"
torch.manual_seed(0)
for (L,C,H,S) in [(2,4,32,16),(1,4,32,16),(3,8,64,16),(2,4,64,16),(3,8,32,16)]:
    feats=[(torch.randn(1,C,S*(l+1),S*(l+1),S*(l+1),device=dev)*0.1).contiguous(memory_format=torch.channels_last_3d) for l in range(L)]
    meta=ops.GridMeta.from_bound([[-1.,1.]]*3)
    lin=[torch.nn.Linear(L*C,H), torch.nn.Linear(H,H), torch.nn.Linear(H,1)]
    ws=[l.weight.data.to(dev) for l in lin]; bs=[l.bias.data.to(dev) for l in lin]
    pack=ops.DecoderPack(ws,bs)
    x=(torch.rand(1000,3,device=dev)*2-1)
    if not ops.sdf_fused_supported(feats, meta, pack): print((L,C,H),"unsupported"); continue
    out,_=ops.sdf_fwd_raw(x,feats,meta,pack,True)
    ref=ops._mlp_torch(ops.encode(x,feats,meta),ws,bs)
    print((L,C,H), "max err", (out-ref).abs().max().item())
