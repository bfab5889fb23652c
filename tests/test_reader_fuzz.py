"""Robustness of the native parsers (csrc/io_native.cpp) on corrupted input: every call must
return (a result or a clean error), never crash or read out of bounds.  Mutations: random byte
flips, truncations, random garbage, length-field tampering."""
import io

import numpy as np
import pytest

from cap2det_amd._lib import Cap2DetHipError
from cap2det_amd.readers import tfrecord as T


def _jpeg(rng, h, w, **kw):
  from PIL import Image
  img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
  b = io.BytesIO()
  Image.fromarray(img).save(b, format="JPEG", **kw)
  return b.getvalue()


def _mutations(rng, data, n):
  data = bytearray(data)
  for _ in range(n):
    d = bytearray(data)
    kind = rng.integers(4)
    if kind == 0:
      for _ in range(int(rng.integers(1, 8))):
        d[int(rng.integers(len(d)))] = int(rng.integers(256))
    elif kind == 1:
      d = d[:int(rng.integers(1, len(d)))]
    elif kind == 2:
      p = int(rng.integers(len(d)))
      d[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
    else:
      p = int(rng.integers(len(d)))
      d[p] = 0xff
      if p + 1 < len(d):
        d[p + 1] = int(rng.choice([0xc0, 0xc4, 0xda, 0xdb, 0xdd, 0xd9, 0xd0, 0x00]))
    yield bytes(d)


def test_jpeg_decoder_survives_corruption():
  rng = np.random.default_rng(0)
  ok = 0
  for kw in (dict(quality=80), dict(quality=60, subsampling=0), dict(quality=90, subsampling=1),
             dict(quality=85, restart_marker_blocks=2)):
    data = _jpeg(rng, 41, 57, **kw)
    for bad in _mutations(rng, data, 400):
      try:
        out = T.decode_jpeg(bad)
        assert out.ndim == 3 and out.shape[2] == 3 and out.dtype == np.uint8
        ok += 1
      except (T.DataError, Cap2DetHipError, MemoryError, ValueError):
        pass
  assert ok > 0          # (flips inside the entropy-coded data usually still decode)
  for n in (0, 1, 2, 3, 10):
    with pytest.raises((T.DataError, Cap2DetHipError)):
      T.decode_jpeg(bytes(rng.integers(0, 256, n, dtype=np.uint8)))


def test_example_parser_survives_corruption():
  rng = np.random.default_rng(1)
  rec = T.encode_example({
      "image/source_id": (T.BYTES, [b"000001"]),
      "image/encoded": (T.BYTES, [bytes(rng.integers(0, 256, 300, dtype=np.uint8))]),
      "image/proposal/bbox/ymin": (T.FLOAT, rng.uniform(0, 1, 50).tolist()),
      "image/object/class/label": (T.INT64, [1, 2, 3, 1 << 40]),
      "image/caption/string": (T.BYTES, [b"a", b"b", b""])})
  keys = ["image/source_id", "image/encoded", "image/proposal/bbox/ymin",
          "image/object/class/label", "image/caption/string", "x"]
  for bad in _mutations(rng, rec, 3000):
    try:
      out = T.parse_example(bad, keys)
      for k, v in out.items():
        assert v is None or isinstance(v, (list, np.ndarray))
    except (T.DataError, Cap2DetHipError):
      pass
  for n in (0, 1, 5, 64):
    junk = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    try:
      T.parse_example(junk, keys)
    except (T.DataError, Cap2DetHipError):
      pass


def test_tfrecord_reader_survives_corruption(tmp_path):
  rng = np.random.default_rng(2)
  path = str(tmp_path / "f.record")
  T.write_records(path, [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (10, 200, 0, 33)])
  blob = open(path, "rb").read()
  for bad in _mutations(rng, blob, 500):
    open(path, "wb").write(bad)
    for verify in (True, False):
      try:
        recs = list(T.iterate_records(path, verify_crc=verify))
        assert all(isinstance(r, bytes) for r in recs)
      except (T.DataError, Cap2DetHipError):
        pass


def test_sanitizer_findings_stay_fixed(tmp_path):
  """Two inputs the ASan / UBSan harness (tools/fuzz_io.cpp) found: a record file ending 12-15
  bytes after a record (the remaining-length subtraction went negative) and a DHT segment whose
  code lengths over-subscribe the code space (lookahead-table index past its end)."""
  rng = np.random.default_rng(3)
  path = str(tmp_path / "f.record")
  T.write_records(path, [b"payload"])
  blob = open(path, "rb").read()
  for extra in range(1, 16):
    open(path, "wb").write(blob + bytes([255] * extra))
    for verify in (True, False):
      with pytest.raises((T.DataError, Cap2DetHipError)):
        list(T.iterate_records(path, verify_crc=verify))
  data = bytearray(_jpeg(rng, 24, 24, quality=80))
  p = data.find(b"\xff\xc4")
  assert p > 0
  data[p + 5] = 3            # three codes of length 1
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(bytes(data))


def test_progressive_jpeg_survives_corruption():
  rng = np.random.default_rng(4)
  ok = 0
  for kw in (dict(quality=80), dict(quality=60, subsampling=0), dict(quality=85, restart_marker_blocks=2)):
    data = _jpeg(rng, 41, 57, progressive=True, **kw)
    for bad in _mutations(rng, data, 400):
      try:
        out = T.decode_jpeg(bad)
        assert out.shape == (41, 57, 3) or out.ndim == 3
        ok += 1
      except (T.DataError, Cap2DetHipError, MemoryError, ValueError):
        pass
  assert ok > 0
