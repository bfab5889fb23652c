"""The C-ABI shared library loads without a GPU and exports exactly what include/*.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from cap2det_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exported():
  out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
  return set(re.findall(r" T (c2d_\w+)", out))


def test_library_exports_every_declared_symbol_and_nothing_else():
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  declared = set(_lib.header_signatures())
  assert declared == _exported()
  lib = _lib.load()
  # the version moves with the ABI: ENTRY_POINTS is the number of entry points at the version the
  # header declares, so adding one without bumping C2D_ABI_VERSION (and this table) fails here
  ENTRY_POINTS = {400: 113, 500: 120, 600: 136}
  version = _lib.header_abi_version()
  assert lib.c2d_version() == version
  assert version in ENTRY_POINTS, "bump tests/test_abi.py with C2D_ABI_VERSION"
  assert len(declared) == ENTRY_POINTS[version], (len(declared), "entry points: bump C2D_ABI_VERSION")
  assert lib.c2d_error_string(0) == b"ok"
  assert lib.c2d_error_string(-1) == b"invalid argument"


def test_header_has_no_torch_or_cpp_types():
  text = open(_lib.HEADER_PATH).read()
  assert 'extern "C"' in text
  code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  for banned in ("torch", "at::", "std::", "hipStream_t", "template", "class "):
    assert banned not in code
  # every entry point cites the reference interface it replaces
  assert text.count("models/") >= 10


def test_invalid_arguments_are_rejected_before_any_launch():
  """Argument validation happens on the host before a kernel is enqueued (no GPU needed)."""
  lib = _lib.load()
  null = None
  assert lib.c2d_conv_fwd(null, 0, 0, null, null, null, null, 0, 0, 1, 7, 7, 16, 16, 1, 1, 1, 1,
                          null) == -1
  assert lib.c2d_crop_and_resize_fwd(null, null, null, null, 1, 4, 4, 16, 1, 14, null) == -1
  assert lib.c2d_adagrad_step(null, null, null, 10, 0.1, 0.0, 1.0, 1.0, null) == -1
  with pytest.raises(_lib.Cap2DetHipError):
    _lib.call("c2d_pool3x3_fwd", null, 0, 0, null, 0, 0, null, 1, 4, 4, 16, 1, 0, null)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  monkeypatch.setattr(_lib, "_lib", None)
  monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
  with pytest.raises(ImportError):
    _lib.load()


def test_crop_backward_shape_query_matches_the_documented_limits():
  """c2d_roi_crop_pool_bwd_ws_shape_supported (round 5): the host-side query a caller
  caches per shape states every limit of the row-owner ROI-crop backward, and FrcnnEngine asks IT —
  not the per-map query — before it commits a shape to the strip kernels."""
  import types
  import torch
  from cap2det_amd import hip_ops as ops
  from cap2det_amd.models.frcnn_engine import FrcnnEngine
  q = ops.roi_crop_pool_bwd_ws_shape_supported
  # benchmark shape: one 32x32x576 map, 2000 boxes, fp32 and bf16 gradients
  assert q(1, 32, 32, 576, 2000, 14, 2, 2, 4) == ops.roi_crop_pool_bwd_ws_supported(32, 576, 14, 2, 2) > 0
  assert q(1, 32, 32, 576, 2000, 14, 2, 2, 2) > 0
  # pooled gradient >= 2 GiB: fp32 7x7x576 cells -> 19,022 boxes; bf16 twice that
  assert q(1, 32, 32, 576, 19021, 14, 2, 2, 4) > 0 and q(1, 32, 32, 576, 19022, 14, 2, 2, 4) == 0
  assert q(1, 32, 32, 576, 19022, 14, 2, 2, 2) > 0 and q(1, 32, 32, 576, 38044, 14, 2, 2, 2) == 0
  # the plan kernel's row table: batch * hf * strips-per-row <= 4095 (84-wide maps: 3 strips)
  assert q(21, 63, 84, 576, 21 * 500, 14, 2, 2, 4) > 0 and q(22, 63, 84, 576, 22 * 500, 14, 2, 2, 4) == 0
  # per-map rules still apply
  assert q(1, 32, 256, 576, 100, 14, 2, 2, 4) == 0 and q(1, 32, 32, 576, 100, 14, 3, 2, 4) == 0
  # the engine's per-shape decision follows the call-level query
  eng = types.SimpleNamespace(crop=14, pool_k=2, pool_s=2)
  pooled = types.SimpleNamespace(t=torch.empty(1, dtype=torch.float32))
  ok = dict(b=21, n=500, fh=63, fw=84, pooled=pooled)
  bad = dict(b=22, n=500, fh=63, fw=84, pooled=pooled)
  assert FrcnnEngine._crop_bwd_ws_ok(eng, ok, 576) is True
  assert FrcnnEngine._crop_bwd_ws_ok(eng, bad, 576) is False
