"""bf16 storage / fp32 accumulate path (BASELINE.json configs[2] and [4]).  The reference has no
reduced-precision mode, so the oracle is the float64 restatement applied to the SAME bf16-rounded
operands; what is left is the fp32 accumulation order (~1e-6 relative) and ONE round-to-nearest
bf16 of the result (relative 2^-9 = 0.2 %).  Tolerance: |got - want| <= 2^-8 |want| + 1e-3 scale."""
import numpy as np
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(a):
  """numpy fp32 -> (bf16 device tensor, the exactly representable fp64 values)."""
  t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV).to(torch.bfloat16)
  return t, t.float().cpu().numpy().astype(np.float64)


def _check(got_bf16, want, what):
  got = got_bf16.float().cpu().numpy().astype(np.float64)
  scale = np.abs(want).max()
  err = np.abs(got - want)
  assert (err <= 2.0 ** -8 * np.abs(want) + 1e-3 * scale).all(), \
      "%s: max err %.3e at scale %.3e" % (what, err.max(), scale)


CASES = [  # n, ih, iw, cin, cout, k, stride
    (3, 7, 7, 32, 64, 1, 1), (5, 7, 7, 48, 96, 3, 1), (5, 7, 7, 32, 64, 3, 2),
    (300, 7, 7, 64, 160, 3, 1), (70, 4, 4, 48, 96, 3, 1), (100, 7, 7, 32, 64, 3, 2),
    (65, 4, 4, 32, 192, 3, 1), (900, 7, 7, 64, 96, 1, 1), (40, 4, 4, 80, 32, 1, 1)]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_bf16(case):
  from cap2det_amd import hip_ops as ops
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(sum(case))
  x, x64 = _bf(rng.standard_normal((n, ih, iw, cin)))
  w, w64 = _bf(rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin))
  scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
  shift = (0.1 * rng.standard_normal(cout)).astype(np.float32)
  c = ref_ops.conv2d(x64, w64, s)
  want = np.maximum(c * scale + shift, 0)
  oh, ow = want.shape[1:3]
  # input / output live in wider concat buffers (channel slices), as in the engine
  ldx, xoff, ldy, yoff = cin + 16, 8, cout + 32, 16
  xb = torch.zeros(n, ih, iw, ldx, device=DEV, dtype=torch.bfloat16); xb[..., xoff:xoff + cin] = x
  yb = torch.full((n, oh, ow, ldy), -7.0, device=DEV, dtype=torch.bfloat16)
  wt = w.permute(0, 1, 3, 2).contiguous().view(k * k, cout, cin)        # [tap][cout][cin]
  ops.conv_fwd(xb, ldx, xoff, wt, torch.from_numpy(scale).to(DEV), torch.from_numpy(shift).to(DEV), yb,
               ldy, yoff, n, ih, iw, cin, cout, k, k, s, True)
  _check(yb[..., yoff:yoff + cout], want, "fwd")
  assert float(yb[..., :yoff].float().max()) == -7.0 and float(yb[..., yoff + cout:].float().min()) == -7.0
  # dgrad (accumulating into an existing bf16 gradient)
  dc, dc64 = _bf(rng.standard_normal((n, oh, ow, cout)))
  want_dx, _ = ref_ops.conv2d_backward(x64, w64, dc64, s)
  base, base64 = _bf(0.5 * rng.standard_normal((n, ih, iw, cin)))
  dx = base.clone()
  ops.conv_dgrad(dc, cout, 0, w.view(k * k, cin, cout), dx, cin, 0, n, ih, iw, cin, cout, k, k, s, True)
  _check(dx, want_dx + base64, "dgrad+accumulate")
  dx2 = torch.empty_like(dx)
  ops.conv_dgrad(dc, cout, 0, w.view(k * k, cin, cout), dx2, cin, 0, n, ih, iw, cin, cout, k, k, s, False)
  _check(dx2, want_dx, "dgrad")


def test_conv1x1_dgrad_multi_bf16():
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(5)
  rows, cin, couts = 3000, 64, [32, 48, 80]
  dcs, ws, want = [], [], 0
  for co in couts:
    d, d64 = _bf(rng.standard_normal((rows, co)))
    w, w64 = _bf(rng.standard_normal((cin, co)) / np.sqrt(co))
    dcs.append(d); ws.append(w); want = want + d64 @ w64.T
  dx = torch.empty(rows, cin, device=DEV, dtype=torch.bfloat16)
  ops.conv1x1_dgrad_multi(dcs, couts, [0, 0, 0], ws, couts, dx, cin, 0, rows, cin, False)
  _check(dx, want, "multi-segment dgrad")
