"""Randomised cross-check of the fused entry points against the unfused launches they replace
(GPU box; the unfused ops are the ones the oracle tests pin):
  c2d_conv_dgrad_bn_relu           == c2d_conv_dgrad + c2d_bn_relu_bwd_partial
  c2d_conv1x1_fwd_multi            == one c2d_conv_fwd per output
  c2d_conv1x1_dgrad_multi_bn_relu  == c2d_conv1x1_dgrad_multi + per-producer c2d_bn_relu_bwd_partial
  python tools/fuzz_fused.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cap2det_amd import hip_ops as ops
DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV).to(dt)
bad = 0
def close(a, b, tol, what, info):
  global bad
  a, b = a.double(), b.double()
  err = float((a - b).abs().max()); sc = float(b.abs().max()) + 1e-30
  if not err <= tol * sc:
    bad += 1
    print("MISMATCH %s %s: err %.3e scale %.3e" % (what, info, err, sc))
for it in range(cases):
  dt = torch.bfloat16 if rng.integers(2) else torch.float32
  al = 8 if dt == torch.bfloat16 else 4
  tol = 2.0 ** -7 if dt == torch.bfloat16 else 2e-5
  # ---- fused input gradient -------------------------------------------------------------
  ih = int(rng.choice([4, 7, 5])); k = int(rng.choice([1, 3])); s = int(rng.choice([1, 2])) if k == 3 else 1
  n = int(rng.choice([3, 70, 130, 300])); cin = int(rng.integers(2, 24)) * al; cout = int(rng.integers(1, 12)) * 16   # (dgrad: K = cout % 16 == 0)
  oh = -(-ih // s)
  w = t(rng.standard_normal((k * k, cin, cout)) / np.sqrt(k * k * cin), dt)
  dc = t(rng.standard_normal((n * oh * oh, cout)), dt)
  y = t(np.maximum(rng.standard_normal((n * ih * ih, cin)), 0), dt)
  scale, beta, gamma = t(rng.uniform(0.5, 1.5, cin)), t(0.1 * rng.standard_normal(cin)), t(rng.uniform(0.5, 1.5, cin))
  info = (str(dt)[6:], n, ih, cin, cout, k, s)
  nb = ops.conv_dgrad_bn_relu_blocks(dt, n, ih, ih, cin, cout, k, k, s)
  out = torch.empty(n * ih * ih, cin, device=DEV, dtype=dt); part = torch.zeros(nb, 2, cin, device=DEV)
  ops.conv_dgrad_bn_relu(dc, cout, 0, w, y, cin, 0, scale, beta, gamma, out, part, n, ih, ih, cin, cout, k, k, s)
  dx = torch.empty(n * ih * ih, cin, device=DEV, dtype=torch.float32)      # reference in fp32 storage
  ops.conv_dgrad(dc.float(), cout, 0, w.float(), dx, cin, 0, n, ih, ih, cin, cout, k, k, s, False)
  dz = dx * (y.float() > 0)
  close(out.float(), dz * scale, tol, "dgrad_bn_relu dc", info)
  close(part.sum(0)[0], dz.sum(0), 1e-3, "dgrad_bn_relu dbeta", info)
  close(part.sum(0)[1], (dz * (y.float() - beta) / gamma).sum(0), 1e-3, "dgrad_bn_relu dgamma", info)
  # ---- several 1x1 convolutions as one GEMM ---------------------------------------------
  rows = int(rng.choice([17000, 20000, 33000])); cin = int(rng.integers(1, 12)) * 16
  nout = int(rng.integers(2, 5)); couts = [int(rng.integers(1, 40)) * al for _ in range(nout)]
  x = t(rng.standard_normal((rows, cin)), dt)
  flat = t(rng.standard_normal(sum(couts) * cin) / np.sqrt(cin), dt)
  outs, wants, off = [], [], 0
  for i, c in enumerate(couts):
    wt = flat[off:off + c * cin].view(1, c, cin); off += c * cin
    sc_, sh_ = t(rng.uniform(0.5, 1.5, c)), t(0.1 * rng.standard_normal(c))
    relu = bool(rng.integers(2))
    want = torch.empty(rows, c, device=DEV, dtype=dt); got = torch.empty(rows, c, device=DEV, dtype=dt)
    ops.conv_fwd(x, cin, 0, wt, sc_, sh_, want, c, 0, rows, 1, 1, cin, c, 1, 1, 1, relu)
    outs.append((wt, sc_, sh_, got, c, 0, c, relu)); wants.append(want)
  ops.conv1x1_fwd_multi(x, cin, 0, ops.conv_outs(outs), rows, cin)
  for o, want in zip(outs, wants):
    if not torch.equal(o[3], want):
      close(o[3].float(), want.float(), tol, "fwd_multi (not bitwise)", (str(dt)[6:], rows, cin, couts))
  # ---- block-boundary fusion (fp32) -----------------------------------------------------
  rows = int(rng.choice([2000, 5000])); nprod = int(rng.integers(2, 5))
  widths = [int(rng.integers(1, 20)) * 8 for _ in range(nprod)]; cin = sum(widths)
  nseg = int(rng.integers(2, 5)); couts = [int(rng.integers(1, 8)) * 16 for _ in range(nseg)]
  dcs = [t(rng.standard_normal((rows, c))) for c in couts]
  ws = [t(rng.standard_normal((cin, c)) / np.sqrt(c)) for c in couts]
  yb = t(np.maximum(rng.standard_normal((rows, cin)), 0)); base = t(rng.standard_normal((rows, cin)))
  acc = bool(rng.integers(2))
  ref = base.clone() if acc else torch.empty(rows, cin, device=DEV)
  ops.conv1x1_dgrad_multi(dcs, couts, [0] * nseg, ws, couts, ref, cin, 0, rows, cin, acc)
  prods, keep, want, sums, o = [], [], ref.clone(), torch.zeros(2, cin, device=DEV), 0
  for i, wd in enumerate(widths):
    if nprod > 2 and i == 1:
      prods.append((None, None, None, wd))
    else:
      sc_, be_, ga_ = t(rng.uniform(0.5, 1.5, wd)), t(0.1 * rng.standard_normal(wd)), t(rng.uniform(0.5, 1.5, wd))
      keep.append((sc_, be_, ga_)); prods.append((sc_, be_, ga_, wd))
      dz = ref[:, o:o + wd] * (yb[:, o:o + wd] > 0)
      want[:, o:o + wd] = dz * sc_; sums[0, o:o + wd] = dz.sum(0)
      sums[1, o:o + wd] = (dz * (yb[:, o:o + wd] - be_) / ga_).sum(0)
    o += wd
  nb = ops.conv1x1_dgrad_multi_bn_relu_blocks(couts, rows, cin)
  out = base.clone() if acc else torch.empty(rows, cin, device=DEV); part = torch.zeros(nb, 2, cin, device=DEV)
  ops.conv1x1_dgrad_multi_bn_relu(dcs, couts, [0] * nseg, ws, couts, yb, cin, 0, ops.bn_producers(prods), out, cin, 0,
                                  part, rows, cin, acc)
  info = (rows, widths, couts, acc)
  close(out, want, 2e-5, "boundary dc", info); close(part.sum(0), sums, 1e-3, "boundary sums", info)
torch.cuda.synchronize()
print("fuzz_fused: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
