import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from miso_amd import ops
dev=torch.device("cuda",0)
step,data=bench.build_workload(dev,0)
for _ in range(3): step.run()
torch.cuda.synchronize()
feats,meta,pack,mask=step.features,step.meta,step.pack,step._mask
sb=ops.SortedBatch(step.n,dev,tiles=step.tiles).sort(step.x,meta)
tk=bench.time_kernel
print("fwd only          ", tk(lambda: ops.sdf_fwd_raw(step.x,feats,meta,pack,True,out=step.sdf,mask=mask,sorted_batch=sb)))
print("fwd+loss (sdf out)", tk(lambda: ops.sdf_fwd_loss_raw(feats,meta,pack,sb,step.aux,mask,step.gpred,step.loss_slots,"L1",1.0,0.0,0.0,sdf_out=step.sdf)))
print("fwd+loss (no sdf) ", tk(lambda: ops.sdf_fwd_loss_raw(feats,meta,pack,sb,step.aux,mask,step.gpred,step.loss_slots,"L1",1.0,0.0,0.0,sdf_out=None)))
