import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops
dev="cuda:0"; torch.manual_seed(0)
hf=wf=32; D=576; N=2000
feat=torch.relu(torch.randn(1,hf,wf,D,device=dev))
c=torch.rand(N,2,device=dev); s=torch.exp(torch.rand(N,2,device=dev)*3.2-3.2)
boxes=torch.cat([(c-s/2).clamp(0,1),(c+s/2).clamp(0,1)],1).contiguous()
ind=torch.zeros(N,dtype=torch.int32,device=dev)
out,arg=ops.roi_crop_pool_fwd(feat,boxes,ind,14,2,2)
dout=torch.randn_like(out); dfeat=torch.zeros_like(feat)
def t(fn,it=20):
    for _ in range(3): fn()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
tf=t(lambda: ops.roi_crop_pool_fwd(feat,boxes,ind,14,2,2,out=out,argmax=arg))
tb=t(lambda: ops.roi_crop_pool_bwd(dout,arg,boxes,ind,dfeat,14,2,2))
ws=torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(1,hf,wf,D,N,14,2,2),dtype=torch.uint8,device=dev)
d1=torch.zeros_like(feat); ops.roi_crop_pool_bwd(dout,arg,boxes,ind,d1,14,2,2)
d2=torch.zeros_like(feat); ops.roi_crop_pool_bwd_ws(dout,arg,boxes,ind,d2,14,2,2,ws)
d3=torch.zeros_like(feat); ops.roi_crop_pool_bwd_ws(dout,arg,boxes,ind,d3,14,2,2,ws)
print("ws vs atomic maxdiff %.3e (scale %.3e); ws deterministic: %s" % ((d1-d2).abs().max().item(), d1.abs().max().item(), bool((d2==d3).all())))
tw=t(lambda: ops.roi_crop_pool_bwd_ws(dout,arg,boxes,ind,dfeat,14,2,2,ws))
print("bwd_ws %.1f us" % (tw*1e3))
byts=4.0*(N*49*D+feat.numel()+boxes.numel())
print("fwd %.1f us %.0f GB/s | bwd %.1f us" % (tf*1e3, byts/tf/1e6, tb*1e3))
