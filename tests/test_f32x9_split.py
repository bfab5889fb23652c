"""The arithmetic identity behind the f32x9 GEMMs, on the CPU restatement (oracle/ref_split.py):
x == hi + mid + lo bit for bit, the planes' low halves are empty, products of planes are exact in
fp32.  The GPU twin (tests/test_gpu_f32x9.py) checks c2d_split3_bf16 against this restatement bit
for bit on 10^7 patterns."""
import numpy as np

from oracle import ref_split


def _patterns(n, seed):
  rng = np.random.default_rng(seed)
  u = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
  special = np.array([0x00000000, 0x80000000, 0x00000001, 0x807fffff, 0x00800000, 0x7f7fffff,
                      0xff7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001, 0x3f800000,
                      0x3f7fffff, 0x3f80ffff, 0x3f8000ff, 0x0affffff, 0x0b000000], np.uint32)
  return np.concatenate([special, u]).view(np.float32)


def test_split_is_exact():
  x = _patterns(2_000_000, 1)
  hi, mid, lo = ref_split.split3(x)
  u = x.view(np.uint32)
  finite = (u & 0x7f800000) != 0x7f800000
  s = ref_split.join3(hi, mid, lo)
  big = finite & (np.abs(x.astype(np.float64)) >= 2.0 ** -110)
  assert big.sum() > 1_000_000
  assert np.array_equal(s[big], x[big].astype(np.float64))
  small = finite & ~big            # lo (and mid) reach into bf16's subnormal range: 2^-133 spacing
  assert small.sum() > 1000
  assert np.all(np.abs(s[small] - x[small].astype(np.float64)) < 2.0 ** -133)
  # every plane of a finite value is finite, of the sign of x or zero, and ordered in magnitude
  for p in (hi, mid, lo):
    assert np.all(np.isfinite(ref_split.bf16_to_f64(p[finite])))
    assert np.all(ref_split.bf16_to_f64(p[finite]) * x[finite].astype(np.float64) >= 0)
  assert np.all(np.abs(ref_split.bf16_to_f64(mid[finite])) <= np.abs(ref_split.bf16_to_f64(hi[finite])) * 2.0 ** -7)
  # Inf / NaN: hi carries the value, the other planes are zero
  bad = ~finite
  assert bad.sum() >= 4
  assert np.array_equal(hi[bad], (u[bad] >> 16).astype(np.uint16))
  assert not mid[bad].any() and not lo[bad].any()


def test_register_form_propagates_non_finite():
  x = np.array([np.inf, -np.inf, np.nan, 1.5], np.float32)
  hi, mid, lo = ref_split.split3(x, finite_planes_only=False)
  s = ref_split.join3(hi, mid, lo)
  assert not np.isfinite(s[:3]).any() and s[3] == 1.5


def test_partial_products_are_exact_in_fp32():
  rng = np.random.default_rng(2)
  x = (rng.standard_normal(200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
  y = (rng.standard_normal(200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
  px = [ref_split.bf16_to_f64(p) for p in ref_split.split3(x)]
  py = [ref_split.bf16_to_f64(p) for p in ref_split.split3(y)]
  total = np.zeros_like(px[0])
  for a in px:
    for b in py:
      prod = a * b                                  # float64: exact
      assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)   # fits fp32
      total += prod
  assert np.array_equal(total, x.astype(np.float64) * y.astype(np.float64))
