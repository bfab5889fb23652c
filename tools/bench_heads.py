"""The fused heads GEMM (1024 -> 112 columns, N = 2000 rows: forward, input gradient, filter
gradient), each timed alone with the instance it dispatches.

  python tools/bench_heads.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cap2det_amd import hip_ops as ops
dev="cuda:0"
def timeit(fn, iters=50):
  for _ in range(3): fn()
  s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e)/iters*1e3
n,cin,cout=2000,1024,int(os.environ.get('HEADS_COLS','112'))
x=torch.randn(n,cin,device=dev); wt=torch.randn(1,cout,cin,device=dev)*0.03; w=wt.permute(0,2,1).contiguous()
y=torch.empty(n,cout,device=dev); dy=torch.randn(n,cout,device=dev); dx=torch.empty(n,cin,device=dev); dw=torch.zeros(1,cin,cout,device=dev)
print("fwd %.1f us"%timeit(lambda: ops.conv_fwd(x,cin,0,wt,None,None,y,cout,0,n,1,1,cin,cout,1,1,1,False)), ops.last_dispatch())
print("dgrad %.1f us"%timeit(lambda: ops.conv_dgrad(dy,cout,0,w,dx,cin,0,n,1,1,cin,cout,1,1,1,False)), ops.last_dispatch())
print("wgrad %.1f us"%timeit(lambda: ops.conv_wgrad(x,cin,0,dy,cout,0,dw,n,1,1,cin,cout,1,1,1)), ops.last_dispatch())
