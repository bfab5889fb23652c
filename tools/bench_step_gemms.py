"""Every implicit-GEMM call of ONE training step's second stage (Mixed_5a-c, N = 2000 ROIs) as the
engine issues them: the fused block-entry forward GEMMs (c2d_conv1x1_fwd_multi), the 3x3 / stride-2
convolutions, their input gradients (four parity launches for stride 2) and the multi-segment
block-entry input gradients — each timed ALONE (20 back-to-back launches), with the kernel
instance it dispatched.  Prints per call and per-step totals.

  python tools/bench_step_gemms.py [bf16|fp32|x9] [fwd|dgrad|wgrad|all]

x9: fp32 operands, the weights bound to bf16 planes (c2d_f32x9_bind): the f32x9 kernels.

Tuning hooks (C2D_TUNE=1 ...) select variants; tools/sweep_step_gemms.sh runs them side by side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops  # noqa: E402

dev = "cuda:0"
X9 = len(sys.argv) > 1 and sys.argv[1] == "x9"
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] in ("fp32", "x9")) else torch.bfloat16
_planes = []


def bind(t):
  if X9:
    _planes.append(ops.x9_planes(t))
  return t
WHAT = sys.argv[2] if len(sys.argv) > 2 else "all"
PEAK = 157.3 if DT == torch.float32 else 2500.0
ITERS = int(os.environ.get("C2D_BENCH_ITERS", "20"))
n = int(os.environ.get("C2D_BENCH_ROIS", "2000"))


COLD = os.environ.get("C2D_BENCH_COLD", "0") == "1"
_flush = None


def timeit(fn):
  """C2D_BENCH_COLD=1: every timed launch follows a 768 MiB fill (its operands come from HBM, as
  inside a training step, instead of from the Infinity Cache the previous repetition left warm)."""
  global _flush
  for _ in range(3):
    fn()
  if COLD:
    if _flush is None:
      _flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    tot = 0.0
    for _ in range(max(ITERS // 4, 3)):
      _flush.fill_(1)
      s = torch.cuda.Event(enable_timing=True)
      e = torch.cuda.Event(enable_timing=True)
      s.record()
      fn()
      e.record()
      torch.cuda.synchronize()
      tot += s.elapsed_time(e)
    return tot / max(ITERS // 4, 3)
  s = torch.cuda.Event(enable_timing=True)
  e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(ITERS):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / ITERS


def rnd(*shape):
  return torch.randn(*shape, device=dev).to(DT)


calls = []   # (kind, label, flops, fn)


def add_conv(label, hw, cin, cout, k, st):
  oh = -(-hw // st)
  fl = 2.0 * n * oh * oh * cin * cout * k * k
  x = rnd(n * hw * hw, cin)
  w = (torch.randn(k * k, cin, cout, device=dev) / (k * k * cin) ** 0.5).to(DT)
  wt = bind(w.permute(0, 2, 1).contiguous())
  bind(w)
  y = torch.empty(n * oh * oh, cout, device=dev, dtype=DT)
  dy = rnd(n * oh * oh, cout)
  dx = torch.empty_like(x)
  sc = torch.ones(cout, device=dev)
  sh = torch.zeros(cout, device=dev)
  calls.append(("fwd", label, fl, lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, n, hw, hw, cin,
                                                       cout, k, k, st, True)))
  calls.append(("dgrad", label, fl, lambda: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, hw, hw, cin,
                                                           cout, k, k, st, False)))
  dw = torch.zeros(k * k, cin, cout, device=dev)
  calls.append(("wgrad", label, fl, lambda: ops.conv_wgrad(x, cin, 0, dy, cout, 0, dw, n, hw, hw, cin,
                                                           cout, k, k, st)))


def add_entry(label, hw, cin, couts, accumulate=True):
  rows = n * hw * hw
  x = rnd(rows, cin)
  flat = bind((torch.randn(sum(couts) * cin, device=dev) / cin ** 0.5).to(DT))
  outs, off = [], 0
  dcs, ws = [], []
  wflat = bind((torch.randn(sum(couts) * cin, device=dev) / cin ** 0.5).to(DT))   # (one arena, as the engine's)
  woff = 0
  for c in couts:
    wt = flat[off:off + c * cin].view(1, c, cin)
    off += c * cin
    outs.append((wt, torch.ones(c, device=dev), torch.zeros(c, device=dev),
                 torch.empty(rows, c, device=dev, dtype=DT), c, 0, c, True))
    dcs.append(rnd(rows, c))
    ws.append(wflat[woff:woff + cin * c].view(cin, c))
    woff += cin * c
  arr = ops.conv_outs(outs)
  fl = 2.0 * rows * cin * sum(couts)
  calls.append(("fwd", label + " entry x%d" % len(couts), fl,
                lambda: ops.conv1x1_fwd_multi(x, cin, 0, arr, rows, cin)))
  dx = torch.zeros(rows, cin, device=dev, dtype=DT)
  calls.append(("dgrad", label + " entry x%d" % len(couts), fl,
                lambda: ops.conv1x1_dgrad_multi(dcs, list(couts), [0] * len(couts), ws, list(couts), dx, cin,
                                                0, rows, cin, accumulate)))
  calls[-1][3].keep = (outs, arr)
  calls[-2][3].keep = (outs, arr)
  for c, dc in zip(couts, dcs):          # (the engine issues one filter gradient per entry convolution)
    dw = torch.zeros(1, cin, c, device=dev)
    calls.append(("wgrad", label.split()[0] + " 1x1 %d->%d" % (cin, c), 2.0 * rows * cin * c,
                  lambda dc=dc, dw=dw, c=c: ops.conv_wgrad(x, cin, 0, dc, c, 0, dw, n, hw, hw, cin, c, 1, 1, 1)))


add_entry("5a 576->(128,192) 7x7", 7, 576, (128, 192))
add_conv("5a 3x3s2 128->192", 7, 128, 192, 3, 2)
add_conv("5a 3x3 192->256", 7, 192, 256, 3, 1)
add_conv("5a 3x3s2 256->256", 7, 256, 256, 3, 2)
add_entry("5b 1024->(352,192,160,128) 4x4", 4, 1024, (352, 192, 160, 128))
add_conv("5b 3x3 192->320", 4, 192, 320, 3, 1)
add_conv("5b 3x3 160->224", 4, 160, 224, 3, 1)
add_conv("5b 3x3 224->224", 4, 224, 224, 3, 1)
add_entry("5c 1024->(352,192,192) 4x4", 4, 1024, (352, 192, 192))
add_conv("5c 3x3 192->320", 4, 192, 320, 3, 1)
add_conv("5c 3x3 192->224", 4, 192, 224, 3, 1)
add_conv("5c 3x3 224->224", 4, 224, 224, 3, 1)
add_conv("5c 1x1 1024->128", 4, 1024, 128, 1, 1)

tot = {}
for kind, label, fl, fn in calls:
  if WHAT != "all" and kind != WHAT:
    continue
  t = timeit(fn)
  inst = ops.last_dispatch()
  tot.setdefault(kind, [0.0, 0.0])
  tot[kind][0] += t
  tot[kind][1] += fl
  print("%-5s %-36s %7.1f us %7.1f TF  %s" % (kind, label, t * 1e3, fl / t / 1e9,
                                               ";".join(i.replace("_kernel", "") for i in inst)))
for k_, (t, fl) in tot.items():
  print("per step %s: %.3f ms, %.1f TF (%.3f of %.0f)" % (k_, t, fl / t / 1e9, fl / t / 1e9 / PEAK, PEAK))
