"""The ONE tuning / A-B switchboard: C2D_TUNE="key=value,key=value" (the C++ side parses the same
variable, csrc/api.hip).  Unset in production: every key takes its default.  Python keys are read
when an engine or a launch plan is BUILT, never per step:

  streams=0            every kernel on one stream (separable per-kernel timings: bench.py --serial)
  branch_streams=0     no branch stream for the short branches of an Inception block
  fuse_bn_bwd=0        separate BN/ReLU-backward launches instead of the fused input-gradient epilogues
  commute_avgpool=0    average pool in front of its 1x1 convolution (the reference order)
  first_stage_fp32=1   a bf16 network keeps its single-image tower in fp32
  f32x9=0              an fp32 network keeps every GEMM on the fp32 matrix pipe
"""
import os


def get(key, default=None):
  for item in os.environ.get("C2D_TUNE", "").split(","):
    k, sep, v = item.partition("=")
    if sep and k == key:
      return v
  return default


def on(key, default=True):
  v = get(key)
  return default if v is None else v != "0"
