"""The 1x1 entry-convolution filter gradients of Mixed_5a/5b/5c (N = 2000 ROIs): the separate
c2d_conv_wgrad launches against ONE c2d_conv1x1_wgrad_multi launch per block.

  [C2D_BENCH_COLD=1] python tools/bench_wgrad_multi.py [bf16|fp32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops  # noqa: E402

dev = "cuda:0"
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
COLD = os.environ.get("C2D_BENCH_COLD", "0") == "1"
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev) if COLD else None


def timeit(fn, iters=10):
  for _ in range(2):
    fn()
  tot = 0.0
  for _ in range(iters):
    if COLD:
      flush.fill_(1)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    tot += s.elapsed_time(e)
  return tot / iters * 1e3


for label, hw, cin, couts in [("5a", 7, 576, (128, 192)), ("5b", 4, 1024, (352, 192, 160, 128)),
                              ("5c", 4, 1024, (352, 192, 192))]:
  rows = 2000 * hw * hw
  x = torch.randn(rows, cin, device=dev).to(DT)
  dcs = [torch.randn(rows, c, device=dev).to(DT) for c in couts]
  dws = [torch.zeros(1, cin, c, device=dev) for c in couts]
  def single():
    for dc, dw, c in zip(dcs, dws, couts):
      ops.conv_wgrad(x, cin, 0, dc, c, 0, dw, 2000, hw, hw, cin, c, 1, 1, 1)
  def multi():
    ops.conv1x1_wgrad_multi(x, cin, 0, dcs, list(couts), [0] * len(couts), dws, list(couts), rows, cin)
  t1, t2 = timeit(single), timeit(multi)
  fl = 2.0 * rows * cin * sum(couts)
  print("%s 1x1 entry filter gradients %d -> %s: separate %.1f us (%.0f TF), one launch %.1f us (%.0f TF)  %s"
        % (label, cin, couts, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, ops.last_dispatch()))
