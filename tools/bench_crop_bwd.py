"""The accumulation half of the row-owner ROI-crop backward (c2d_roi_crop_pool_bwd_run: strip kernel
+ ordered sum of the parts) alone, on the benchmark's map / box distribution, fp32 and bf16
gradients; checks determinism and the atomic form's result."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops

dev = "cuda:0"
torch.manual_seed(0)
hf = wf = int(os.environ.get("HW", "32"))
D, N = 576, int(os.environ.get("N", "2000"))
feat = torch.relu(torch.randn(1, hf, wf, D, device=dev))
c = torch.rand(N, 2, device=dev)
s = torch.exp(torch.rand(N, 2, device=dev) * 3.2 - 3.2)
boxes = torch.cat([(c - s / 2).clamp(0, 1), (c + s / 2).clamp(0, 1)], 1).contiguous()
ind = torch.zeros(N, dtype=torch.int32, device=dev)
out, arg = ops.roi_crop_pool_fwd(feat, boxes, ind, 14, 2, 2)
ws = torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(1, hf, wf, D, N, 14, 2, 2), dtype=torch.uint8, device=dev)
ops.roi_crop_pool_bwd_prepare(boxes, ind, 1, hf, wf, D, 14, 2, 2, ws)


def t(fn, it=30):
  for _ in range(3):
    fn()
  a = torch.cuda.Event(enable_timing=True)
  b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(it):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / it * 1e3


for dt in (torch.float32, torch.bfloat16):
  dout = torch.randn(out.shape, device=dev).to(dt)
  d1 = torch.zeros_like(feat)
  ops.roi_crop_pool_bwd(dout.float(), arg, boxes, ind, d1, 14, 2, 2)
  d2 = torch.zeros_like(feat)
  ops.roi_crop_pool_bwd_run(dout, arg, boxes, ind, d2, 14, 2, 2, ws)
  d3 = torch.zeros_like(feat)
  ops.roi_crop_pool_bwd_run(dout, arg, boxes, ind, d3, 14, 2, 2, ws)
  dfeat = torch.zeros_like(feat)
  us = t(lambda: ops.roi_crop_pool_bwd_run(dout, arg, boxes, ind, dfeat, 14, 2, 2, ws))
  print("%s run %.1f us | vs atomic maxdiff %.2e (scale %.2e) deterministic %s checksum %.9e" % (
      str(dt).replace("torch.", ""), us, (d1 - d2).abs().max().item(), d1.abs().max().item(),
      bool((d2 == d3).all()), d2.double().sum().item()))

# list-length balance: counts[hf][kBinSegs] sits behind the two axis tables in the workspace
off = (2 * N * 14 * 16 + 255) // 256 * 256
nr = -(-wf // int(os.environ.get("C2D_ROI_STRIP_COLS", "32") or wf))     # column ranges per row
counts = ws[off:off + hf * nr * 16 * 4].view(torch.int32).view(hf * nr, 16).cpu()
print("entries per (strip row, segment): mean %.0f max %d min %d; per row: %s" % (
    counts.float().mean().item(), counts.max().item(), counts.min().item(), counts.sum(1).tolist()))
print("entries per cell: %.2f" % (counts.sum().item() / (N * 49.0)))
