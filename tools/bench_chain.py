"""How one fp32 implicit-GEMM launch's time depends on workgroups per CU and on the MFMA work per
workgroup: the 3x3 convolution 224 -> cout on 4x4 maps (pixel-major 128x64 tiles, 44 visited slabs),
cout = 32 .. 256 at N = 2000 ROIs (252 m-tiles: one workgroup per CU and n-tile) and N = 4000 / 8000
at cout = 64 / 32.  Round 3 reading: a lone workgroup takes 93 us whether its tile is
full or half (its own chain: 4400 cycles per slab, 2048 of them its MFMAs); 2 / 3 / 4 per CU take
145 / 183 / 237 us (72 % of the matrix pipe at four); four HALF tiles per CU take 144 us; and 224
columns (three full tiles + one half per row block) take the same 237 us as 256.

  python tools/bench_chain.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cap2det_amd import hip_ops as ops
dev="cuda:0"
def timeit(fn, iters=20):
  for _ in range(3): fn()
  s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e)/iters*1e3
for n,cout in [(2000,32),(2000,64),(2000,128),(2000,192),(2000,224),(2000,256),(4000,64),(8000,64),(8000,32)]:
  cin,hw,k=224,4,3
  x=torch.randn(n*hw*hw,cin,device=dev); wt=torch.randn(k*k,cout,cin,device=dev)*0.02
  y=torch.empty(n*hw*hw,cout,device=dev)
  t=timeit(lambda: ops.conv_fwd(x,cin,0,wt,None,None,y,cout,0,n,hw,hw,cin,cout,k,k,1,True))
  fl=2.0*n*hw*hw*cin*cout*k*k
  print("n=%d cout=%d: %.1f us  %.1f TF  %s"%(n,cout,t,fl/t/1e6, ops.last_dispatch()))
