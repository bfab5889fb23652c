"""Host side of the input pipeline — no GPU needed: CRC-32C against the
RFC 3720 vectors, TFRecord framing incl. corruption, the native tf.Example parser against an
independent pure-Python decoder, the native JPEG decoder against the system libjpeg-turbo
(Pillow) bit for bit, and the reference's own known answers for parse_texts
(core/preprocess_test.py:133-170), the resizer shapes (core/imgproc_test.py:198-218) and the box
flip / rescale (core/box_utils_test.py:13-49, tests/golden/box_utils_known_answers.json)."""
import io
import json
import os
import struct

import numpy as np
import pytest

from cap2det_amd.readers import cap2det_reader as reader
from cap2det_amd.readers import tfrecord as T
from oracle import ref_reader as R


def _lib():
  from cap2det_amd import _lib
  return _lib.load()


def test_crc32c_rfc3720_vectors():
  lib = _lib()
  assert lib.c2d_crc32c(b"123456789", 9) == 0xE3069283
  assert lib.c2d_crc32c(bytes(32), 32) == 0x8A9136AA
  assert lib.c2d_crc32c(b"\xff" * 32, 32) == 0x62A8AB43
  assert lib.c2d_crc32c(bytes(range(32)), 32) == 0x46DD794E
  assert lib.c2d_crc32c(bytes(range(31, -1, -1)), 32) == 0x113FDB5C
  rng = np.random.default_rng(0)
  for n in (0, 1, 7, 8, 9, 63, 64, 65, 1000):
    data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    assert lib.c2d_crc32c(data, n) == R.crc32c(data)
    assert lib.c2d_masked_crc32c(data, n) == R.masked_crc(data)


def test_tfrecord_round_trip_and_corruption(tmp_path):
  rng = np.random.default_rng(1)
  payloads = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 1, 100, 70000)]
  path = str(tmp_path / "a.record")
  T.write_records(path, payloads)
  blob = open(path, "rb").read()
  assert R.read_records(blob) == payloads            # independent reader accepts the framing
  assert list(T.iterate_records(path)) == payloads
  assert len(blob) == sum(len(p) + 16 for p in payloads)
  # flipped payload byte, flipped length byte, truncated file
  for pos in (16 + 12 + 16 + 1 + 12 + 5, 16 + 12 + 3):
    bad = bytearray(blob); bad[pos] ^= 0x40
    open(path, "wb").write(bytes(bad))
    with pytest.raises(T.DataError):
      list(T.iterate_records(path))
  open(path, "wb").write(blob[:-3])
  with pytest.raises(T.DataError):
    list(T.iterate_records(path))
  open(path, "wb").write(blob[:-3])
  assert len(list(zip(range(3), T.iterate_records(path)))) == 3   # the intact prefix still reads
  open(path, "wb").write(b"")
  assert list(T.iterate_records(path)) == []


def _unpacked_feature_example():
  """An Example whose float / int64 lists are written UNPACKED (one tag per element), which
  protobuf parsers must accept too."""
  def ld(field, payload):
    return T._varint((field << 3) | 2) + T._varint(len(payload)) + payload
  flist = b"".join(T._varint((1 << 3) | 5) + struct.pack("<f", v) for v in (1.5, -2.0))
  ilist = b"".join(T._varint((1 << 3) | 0) + T._varint(v) for v in (7, 300))
  body = b""
  for name, kind, lst in (("f", 2, flist), ("i", 3, ilist)):
    body += ld(1, ld(1, name.encode()) + ld(2, ld(kind, lst)))
  return ld(1, body)


def test_example_parser_matches_independent_decoder():
  rng = np.random.default_rng(2)
  feats = {
      "image/source_id": (T.BYTES, [b"2008_000123"]),
      "image/encoded": (T.BYTES, [rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()]),
      "image/caption/string": (T.BYTES, [b"a", b"dog", b"", "café".encode()]),
      "image/caption/offset": (T.INT64, [0, 2]),
      "image/caption/length": (T.INT64, [2, 2]),
      "image/proposal/bbox/ymin": (T.FLOAT, rng.uniform(0, 1, 700).astype(np.float32).tolist()),
      "image/object/class/label": (T.INT64, [1, 20, -3, 1 << 50]),
      "empty/floats": (T.FLOAT, []),
      "image/height": (T.INT64, [375]),
  }
  rec = T.encode_example(feats)
  keys = sorted(feats) + ["not/there"]
  got = T.parse_example(rec, keys)
  want = R.parse_example(rec)
  assert got["not/there"] is None
  for k in feats:
    if feats[k][0] == T.BYTES:
      assert got[k] == want[k] == list(feats[k][1])
    else:
      np.testing.assert_array_equal(got[k], want[k])
      np.testing.assert_array_equal(got[k], np.asarray(feats[k][1], got[k].dtype))
  un = _unpacked_feature_example()
  got = T.parse_example(un, ["f", "i"])
  np.testing.assert_array_equal(got["f"], [1.5, -2.0]); np.testing.assert_array_equal(got["i"], [7, 300])
  np.testing.assert_array_equal(R.parse_example(un)["f"], [1.5, -2.0])
  for cut in (1, len(rec) // 2, len(rec) - 1):
    with pytest.raises(T.DataError):
      T.parse_example(rec[:cut], keys)


def _jpeg(img, **kw):
  from PIL import Image
  b = io.BytesIO()
  Image.fromarray(img).save(b, format="JPEG", **kw)
  return b.getvalue()


def _photo_like(rng, h, w):
  y, x = np.mgrid[0:h, 0:w]
  base = np.stack([128 + 100 * np.sin(x / 7.0 + c) * np.cos(y / 11.0 - c) for c in range(3)], -1)
  return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,kw", [
    (16, 16, dict(quality=90, subsampling=0)), (37, 53, dict(quality=90, subsampling=1)),
    (37, 53, dict(quality=75, subsampling=2)), (1, 1, dict(quality=95)), (8, 9, dict(quality=50)),
    (120, 161, dict(quality=30, subsampling=2)), (65, 47, dict(quality=100, subsampling=2)),
    (67, 45, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    (67, 45, dict(quality=85, subsampling=1, restart_marker_rows=1)),
    (333, 500, dict(quality=92, subsampling=2, optimize=True))])
def test_jpeg_decoder_bit_exact_vs_libjpeg_turbo(h, w, kw):
  rng = np.random.default_rng(h * 1000 + w)
  for img in (_photo_like(rng, h, w), rng.integers(0, 256, (h, w, 3)).astype(np.uint8)):
    data = _jpeg(img, **kw)
    np.testing.assert_array_equal(T.decode_jpeg(data), R.decode_jpeg(data))
  gray = _jpeg(_photo_like(rng, h, w)[..., 0], quality=kw.get("quality", 90))
  np.testing.assert_array_equal(T.decode_jpeg(gray), R.decode_jpeg(gray))     # replicated to RGB


@pytest.mark.parametrize("h,w,kw", [
    (16, 16, dict(quality=90, subsampling=0)), (37, 53, dict(quality=90, subsampling=1)),
    (37, 53, dict(quality=75, subsampling=2)), (1, 1, dict(quality=95)), (9, 8, dict(quality=40)),
    (120, 161, dict(quality=30, subsampling=2)), (65, 47, dict(quality=100, subsampling=2)),
    (67, 45, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    (67, 45, dict(quality=85, subsampling=1, restart_marker_rows=1)),
    (333, 500, dict(quality=92, subsampling=2, optimize=True))])
def test_progressive_jpeg_bit_exact_vs_libjpeg_turbo(h, w, kw):
  """SOF2 files (DC / AC first + refinement scans, EOB runs, per-scan Huffman tables, restart
  intervals): tf.image.decode_jpeg of readers/cap2det_reader.py:91-92 takes them; a small share
  of COCO / Flickr30k is stored this way."""
  rng = np.random.default_rng(h * 1000 + w + 1)
  for img in (_photo_like(rng, h, w), rng.integers(0, 256, (h, w, 3)).astype(np.uint8)):
    data = _jpeg(img, progressive=True, **kw)
    assert b"\xff\xc2" in data
    np.testing.assert_array_equal(T.decode_jpeg(data), R.decode_jpeg(data))
  gray = _jpeg(_photo_like(rng, h, w)[..., 0], quality=kw.get("quality", 90), progressive=True)
  np.testing.assert_array_equal(T.decode_jpeg(gray), R.decode_jpeg(gray))


def test_jpeg_decoder_rejects_what_it_does_not_support():
  rng = np.random.default_rng(3)
  from cap2det_amd._lib import Cap2DetHipError
  prog = _jpeg(_photo_like(rng, 40, 40), quality=80, progressive=True)
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(prog[:len(prog) // 3])                 # cut inside the first scans
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(prog[:-2])                             # EOI (and nothing else) missing
  good = _jpeg(_photo_like(rng, 40, 40), quality=80)
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(good[:len(good) // 2])                 # truncated entropy-coded data
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(good[:100])
  with pytest.raises((T.DataError, Cap2DetHipError)):
    T.decode_jpeg(b"not a jpeg at all")


def test_reference_known_answers_for_host_logic():
  # core/preprocess_test.py:133-170
  tokens = ["first", "second", "text", "the", "third", "text"]
  with pytest.raises(ValueError):
    reader.parse_texts(tokens, [0, 1], [1, 2, 3])
  n, strings, lengths = reader.parse_texts(tokens, [0, 1, 3], [1, 2, 3])
  assert n == 3 and list(lengths) == [1, 2, 3]
  assert strings == [["first", "", ""], ["second", "text", ""], ["the", "third", "text"]]
  assert reader.parse_texts([], [], []) == (0, [], pytest.approx(np.zeros(0)))
  # core/imgproc_test.py:198-218 (keep_aspect_ratio_resizer), :142-160 (fixed_shape_resizer)
  from cap2det_amd.protos import image_resizer_pb2, text_format
  opt = image_resizer_pb2.ImageResizer()
  text_format.Merge("keep_aspect_ratio_resizer { min_dimension: 900 }", opt)
  assert reader.resized_shape(opt, 300, 400) == (900, 1200)
  assert reader.resized_shape(opt, 400, 300) == (1200, 900)
  opt = image_resizer_pb2.ImageResizer()
  text_format.Merge("fixed_shape_resizer { height: 600 width: 800 }", opt)
  assert reader.resized_shape(opt, 300, 400) == (600, 800)
  opt = image_resizer_pb2.ImageResizer()
  text_format.Merge("default_resizer { }", opt)
  assert reader.resized_shape(opt, 300, 400) == (300, 400)
  # core/box_utils_test.py:34-49 (flip), :13-32 (scale_to_new_size = the batch box rescale)
  path = os.path.join(os.path.dirname(__file__), "golden", "box_utils_known_answers.json")
  cases = {c["fn"]: c for c in json.load(open(path))["cases"]}
  c = cases["flip_left_right"]
  np.testing.assert_allclose(reader.flip_boxes_left_right(c["box"]), c["expected"])
  np.testing.assert_allclose(R.flip_boxes(np.asarray(c["box"], np.float32)), c["expected"])


def test_hash_bucket_is_stable_and_spread():
  ids = ["%06d" % i for i in range(4000)]
  buckets = np.array([T.to_hash_bucket(s, 8) for s in ids])
  assert np.array_equal(buckets, [T.to_hash_bucket(s.encode(), 8) for s in ids])
  counts = np.bincount(buckets, minlength=8)
  assert counts.min() > 400 and counts.max() < 600          # ~500 each
  # shards are a partition
  assert sum(int((buckets == k).sum()) for k in range(8)) == len(ids)


def test_device_prefetcher_on_cpu():
  """readers/prefetch.py (the reference's dataset.prefetch, readers/cap2det_reader.py:266): order
  kept, the producer's exception re-raised in the consumer, close() ends an endless source."""
  import torch
  from cap2det_amd.readers.prefetch import DevicePrefetcher

  def gen(n, fail_at=None):
    for i in range(n):
      if i == fail_at:
        raise RuntimeError("boom")
      yield {"i": i, "t": torch.zeros(3)}

  p = DevicePrefetcher(gen(7), "cpu", depth=2)
  got = list(p)
  assert [b["i"] for b in got] == list(range(7)) and all(b["_ready"] is None for b in got)
  assert next(p, "end") == "end"
  p.close()
  p = DevicePrefetcher(gen(5, fail_at=2), "cpu")
  seen = []
  try:
    for b in p:
      seen.append(b["i"])
    raise AssertionError("the producer's exception was swallowed")
  except RuntimeError as e:
    assert str(e) == "boom" and seen == [0, 1]
  p.close()
  p = DevicePrefetcher(gen(10 ** 9), "cpu", depth=2)
  next(p)
  p.close()
  assert not p._thread.is_alive()
