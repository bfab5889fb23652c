"""Per-block phase times of ONE bf16 implicit-GEMM launch from the diagnostic build
(make -C cap2det_amd/csrc EXTRA_CXXFLAGS=-DC2D_RING_TRACE): cycles of wave 0 in the prologue, the
K loop, the epilogue, and — inside the loop — at the wait + barrier and at the DMA issue.

  python tools/trace_ring.py fwd|dgrad  n ih cin cout k stride
  python tools/trace_ring.py entryf|entryd  n ih cin cout[,cout...]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import _lib, hip_ops as ops  # noqa: E402

dev = "cuda:0"
DT = torch.float32 if os.environ.get("C2D_TRACE_FP32") else torch.bfloat16
which = sys.argv[1]
n, ih, cin = [int(v) for v in sys.argv[2:5]]
couts = [int(v) for v in sys.argv[5].split(",")]
k, st = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (1, 1)


def rnd(*shape):
  return torch.randn(*shape, device=dev).to(DT)


rows = n * ih * ih
x = rnd(rows, cin)
if which in ("fwd", "dgrad"):
  cout = couts[0]
  oh = -(-ih // st)
  w = (torch.randn(k * k, cin, cout, device=dev) / (k * k * cin) ** 0.5).to(DT)
  wt = w.permute(0, 2, 1).contiguous()
  y = torch.empty(n * oh * oh, cout, device=dev, dtype=DT)
  dy = rnd(n * oh * oh, cout)
  dx = torch.empty_like(x)
  sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
  if which == "fwd":
    run = lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, n, ih, ih, cin, cout, k, k, st, True)
  else:
    run = lambda: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, st, False)
  flops = 2.0 * n * oh * oh * cin * cout * k * k
else:
  flat = (torch.randn(sum(couts) * cin, device=dev) / cin ** 0.5).to(DT)
  outs, off, dcs, ws = [], 0, [], []
  for c in couts:
    wt = flat[off:off + c * cin].view(1, c, cin)
    off += c * cin
    outs.append((wt, torch.ones(c, device=dev), torch.zeros(c, device=dev),
                 torch.empty(rows, c, device=dev, dtype=DT), c, 0, c, True))
    dcs.append(rnd(rows, c))
    ws.append((torch.randn(cin, c, device=dev) / c ** 0.5).to(DT))
  arr = ops.conv_outs(outs)
  dx = torch.zeros(rows, cin, device=dev, dtype=DT)
  if which == "entryf":
    run = lambda: ops.conv1x1_fwd_multi(x, cin, 0, arr, rows, cin)
  else:
    run = lambda: ops.conv1x1_dgrad_multi(dcs, list(couts), [0] * len(couts), ws, list(couts), dx, cin, 0,
                                          rows, cin, True)
  flops = 2.0 * rows * cin * sum(couts)

lib = ctypes.CDLL(_lib.LIB_PATH)
trace = torch.zeros(16 * 65536, dtype=torch.int64, device=dev)
for _ in range(3):
  run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%s: %.1f us per launch, %.1f TF, instance %s" % (" ".join(sys.argv[1:]), ms * 1e3, flops / ms / 1e9,
                                                      ";".join(ops.last_dispatch())))
if not hasattr(lib, "c2d_debug_set_ring_trace"):
  sys.exit(0)      # (product build: the launch time only)
lib.c2d_debug_set_ring_trace(ctypes.c_void_p(trace.data_ptr()))
run()
torch.cuda.synchronize()
lib.c2d_debug_set_ring_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy().reshape(-1, 16)
t = t[t[:, 0] != 0]
t0, pro, loop, epi, wait, issue, cnt, rt = [t[:, i].astype(np.float64) for i in range(8)]
print("blocks %d, stages per block med %d" % (len(t), np.median(cnt)))
tot = pro + loop + epi
med = lambda v: float(np.median(v))
print("cycles of wave 0 per block (median): prologue %.0f  K loop %.0f  epilogue %.0f  total %.0f"
      % (med(pro), med(loop), med(epi), med(tot)))
print("  epilogue: barrier %.0f, BN vectors / routing %.0f, first strip %.0f, other strips %.0f"
      % tuple(med(t[:, i].astype(np.float64)) for i in (8, 9, 10, 11)))
print("  share: prologue %.2f  loop %.2f  epilogue %.2f" % (med(pro / tot), med(loop / tot), med(epi / tot)))
print("per stage (loop / stages): %.0f cycles; of it wait+barrier %.0f, cursor+DMA issue %.0f, frag reads+MFMA %.0f"
      % (med(loop / cnt), med(wait / cnt), med(issue / cnt), med((loop - wait - issue) / cnt)))
span = (rt.max() - rt.min()) / 100.0
print("last-block-end spread %.1f us (s_memrealtime); sum of block cycles / (launch us x 2100) = %.1f resident blocks"
      % (span, tot.sum() / (ms * 1e3 * 2100.0)))
order = np.argsort(t0)
first = order[: max(len(order) // 4, 1)]
last = order[-max(len(order) // 4, 1):]
print("first quarter of blocks: total %.0f (loop/stage %.0f); last quarter: total %.0f (loop/stage %.0f)"
      % (med(tot[first]), med((loop / cnt)[first]), med(tot[last]), med((loop / cnt)[last])))
