"""ROI crop forward variants: with / without the arg-max store, fp32 / bf16 output."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops
dev="cuda:0"; torch.manual_seed(0)
hf=wf=32; D=576; N=2000
feat=torch.relu(torch.randn(1,hf,wf,D,device=dev))
c=torch.rand(N,2,device=dev); s=torch.exp(torch.rand(N,2,device=dev)*3.2-3.2)
boxes=torch.cat([(c-s/2).clamp(0,1),(c+s/2).clamp(0,1)],1).contiguous()
ind=torch.zeros(N,dtype=torch.int32,device=dev)
def t(fn,it=30):
    for _ in range(3): fn()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for dt in (torch.float32, torch.bfloat16):
    out=torch.empty(N,7,7,D,device=dev,dtype=dt); arg=torch.empty(N,7,7,D,device=dev,dtype=torch.uint8)
    es = 4 if dt==torch.float32 else 2
    byts=es*N*49*D+4.0*(feat.numel()+boxes.numel())
    t1=t(lambda: ops.roi_crop_pool_fwd(feat,boxes,ind,14,2,2,out=out,argmax=arg))
    t2=t(lambda: ops.roi_crop_pool_fwd(feat,boxes,ind,14,2,2,out=out,argmax=None,want_argmax=False))
    print("%s: with argmax %.1f us %.0f GB/s | without %.1f us %.0f GB/s" % (dt, t1*1e3, byts/t1/1e6, t2*1e3, byts/t2/1e6))
