"""Forward convolutions of the single-image first stage (Inception-V2 up to Mixed_4e, one 500x500
image), each timed alone in fp32 (igemm_small_kernel) and bf16 (what the bf16 entry point picks).

  python tools/bench_first_stage.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops  # noqa: E402

dev = "cuda:0"
ITERS = 30


def timeit(fn):
  for _ in range(3):
    fn()
  s = torch.cuda.Event(enable_timing=True)
  e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(ITERS):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / ITERS * 1e3


shapes = [  # (label, hw, cin, cout, k, stride)
    ("conv2 1x1 64->64 @125", 125, 64, 64, 1, 1), ("conv2 3x3 64->192 @125", 125, 64, 192, 3, 1),
    ("3b 1x1 192->64 @63", 63, 192, 64, 1, 1), ("3b 3x3 64->64 @63", 63, 64, 64, 3, 1),
    ("3b 3x3 64->96 @63", 63, 64, 96, 3, 1), ("3b 3x3 96->96 @63", 63, 96, 96, 3, 1),
    ("3c 1x1 256->64 @63", 63, 256, 64, 1, 1), ("4a 3x3s2 128->160 @63", 63, 128, 160, 3, 2),
    ("4a 3x3s2 96->96 @63", 63, 96, 96, 3, 2),
    ("4b 1x1 576->224 @32", 32, 576, 224, 1, 1), ("4b 3x3 64->96 @32", 32, 64, 96, 3, 1),
    ("4b 3x3 96->128 @32", 32, 96, 128, 3, 1), ("4b 3x3 128->128 @32", 32, 128, 128, 3, 1),
    ("4d 3x3 128->160 @32", 32, 128, 160, 3, 1), ("4d 3x3 160->192 @32", 32, 160, 192, 3, 1),
    ("4e 3x3 192->256 @32", 32, 192, 256, 3, 1), ("4e 3x3 160->192 @32", 32, 160, 192, 3, 1),
]
tot = {torch.float32: 0.0, torch.bfloat16: 0.0}
for label, hw, cin, cout, k, st in shapes:
  row = []
  for dt in (torch.float32, torch.bfloat16):
    oh = -(-hw // st)
    x = torch.randn(hw * hw, cin, device=dev).to(dt)
    wt = (torch.randn(k * k, cout, cin, device=dev) / (k * k * cin) ** 0.5).to(dt)
    y = torch.empty(oh * oh, cout, device=dev, dtype=dt)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    t = timeit(lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, 1, hw, hw, cin, cout, k, k, st, True))
    tot[dt] += t
    row.append("%6.1f us %s" % (t, ";".join(i.replace("_kernel", "") for i in ops.last_dispatch())))
  print("%-26s fp32 %-48s bf16 %s" % (label, row[0], row[1]))
print("sum: fp32 %.1f us, bf16 %.1f us" % (tot[torch.float32], tot[torch.bfloat16]))
