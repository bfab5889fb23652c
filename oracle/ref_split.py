"""ORACLE (test infrastructure, never shipped, never imported by cap2det_amd/).

numpy restatement of the truncation split behind the f32x9 GEMMs (cap2det_amd/csrc/igemm_x9.hip,
igemm_ring.h): an fp32 value as the sum of three bf16 terms.  Not a reference algorithm — the
reference multiplies fp32 tensors (slim.conv2d, models/utils.py:165-167); this file states the
arithmetic identity the kernels rely on so that tests can check it bit by bit:

  hi  = the top 16 bits of x                      (bf16, truncation: 8 significant bits)
  mid = the top 16 bits of r1 = x - hi            (r1 exact in fp32: <= 16 significant bits)
  lo  = r2 = r1 - mid                             (<= 8 significant bits: its low 16 bits are zero)

x == hi + mid + lo exactly whenever lo is representable, i.e. |x| >= 2^-110 (lo's exponent is at
most 23 below x's and bf16, like fp32, ends at 2^-133 = its smallest subnormal); below that the
sum is within 2^-133 of x.  Every product of two such terms has <= 16 significant bits and is exact
in fp32, so x * y = the fp32-accumulated sum of nine exact partial products.
"""
import numpy as np


def split3(x, finite_planes_only=True):
  """x float32 array -> (hi, mid, lo) uint16 arrays of bf16 bit patterns.

  finite_planes_only: Inf / NaN keep hi and zero mid / lo (c2d_split3_bf16: the weight planes);
  False: the in-register form of the activation fragments, whose mid is the NaN of x - x."""
  x = np.ascontiguousarray(x, np.float32)
  u = x.view(np.uint32)
  with np.errstate(invalid="ignore"):
    hi_f = (u & np.uint32(0xffff0000)).view(np.float32)
    r1 = (x - hi_f).astype(np.float32)
    u1 = r1.view(np.uint32)
    mid_f = (u1 & np.uint32(0xffff0000)).view(np.float32)
    r2 = (r1 - mid_f).astype(np.float32)
  hi = (u >> np.uint32(16)).astype(np.uint16)
  mid = (u1 >> np.uint32(16)).astype(np.uint16)
  lo = (r2.view(np.uint32) >> np.uint32(16)).astype(np.uint16)
  if finite_planes_only:
    bad = (u & np.uint32(0x7f800000)) == np.uint32(0x7f800000)
    mid = np.where(bad, np.uint16(0), mid)
    lo = np.where(bad, np.uint16(0), lo)
  return hi, mid, lo


def bf16_to_f64(bits):
  return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32).astype(np.float64)


def join3(hi, mid, lo):
  """float64 sum of the three planes (exact: 24 bits over at most 39 binades)."""
  return bf16_to_f64(hi) + bf16_to_f64(mid) + bf16_to_f64(lo)
