"""TFRecord files of tf.Example protos: iteration, parsing and (for fixtures / dataset tools)
writing, over the native routines of csrc/io_native.cpp.

Reference: the files are produced by dataset-tools/create_pascal_tf_record.py:147-196 and
create_coco_tf_record.py:197-242 (`tf.python_io.TFRecordWriter`, `tf.train.Example`) and consumed
by readers/cap2det_reader.py:40-59,217-218 (`tf.data.TFRecordDataset`,
`tf.parse_single_example`).
"""
import ctypes
import struct

import numpy as np

from cap2det_amd import _lib

BYTES, FLOAT, INT64 = 1, 2, 3


class DataError(ValueError):
  """Malformed record / proto / image data."""


def _check(rc, what):
  if rc == -5:
    raise DataError(what + ": malformed data")
  _lib.check(rc, what)


def iterate_records(path, verify_crc=True):
  """Yields the payload bytes of every record of one TFRecord file (memory mapped)."""
  import os
  lib = _lib.load()
  size = os.path.getsize(path)
  if size == 0:
    return
  arr = np.memmap(path, dtype=np.uint8, mode="r")
  base = arr.ctypes.data
  off, length = ctypes.c_longlong(), ctypes.c_longlong()
  pos = 0
  while True:
    nxt = lib.c2d_tfrecord_next(base, size, pos, ctypes.byref(off), ctypes.byref(length),
                                int(verify_crc))
    if nxt == 0:
      break
    if nxt < 0:
      _check(int(nxt), "TFRecord %s @%d" % (path, pos))
    yield bytes(arr[off.value:off.value + length.value])
    pos = nxt


def frame_record(payload):
  lib = _lib.load()
  out = np.empty(len(payload) + 16, np.uint8)
  n = lib.c2d_tfrecord_frame(payload, len(payload), out.ctypes.data)
  assert n == len(payload) + 16
  return out.tobytes()


def write_records(path, payloads):
  with open(path, "wb") as f:
    for p in payloads:
      f.write(frame_record(p))


def parse_example(record, keys):
  """tf.parse_single_example for the given feature names.  Returns {key: value} with bytes
  features as lists of bytes, float features as float32 arrays, int64 features as int64 arrays;
  absent features map to None."""
  lib = _lib.load()
  n = len(record)
  nk = len(keys)
  ckeys = (ctypes.c_char_p * nk)(*[k.encode() for k in keys])
  kinds = (ctypes.c_int * nk)()
  counts = (ctypes.c_longlong * nk)()
  starts = (ctypes.c_longlong * nk)()
  cap = max(n // 4 + 8, 16)            # no feature list can hold more values than bytes / 1
  floats = np.empty(cap, np.float32)
  ints = np.empty(n + 8, np.int64)
  spans = np.empty(2 * (n // 2 + 8), np.int64)
  rc = lib.c2d_example_parse(record, n, ckeys, nk, kinds, counts, starts, floats.ctypes.data, cap,
                             ints.ctypes.data, n + 8, spans.ctypes.data, n // 2 + 8)
  _check(rc, "tf.Example")
  out = {}
  for i, k in enumerate(keys):
    s, c = starts[i], counts[i]
    if kinds[i] == BYTES:
      out[k] = [record[spans[2 * j]:spans[2 * j] + spans[2 * j + 1]] for j in range(s, s + c)]
    elif kinds[i] == FLOAT:
      out[k] = floats[s:s + c].copy()
    elif kinds[i] == INT64:
      out[k] = ints[s:s + c].copy()
    else:
      out[k] = None
  return out


# -- tf.Example encoding (protobuf wire format; packed repeated scalars as TF writes them) ------

def _varint(v):
  v &= (1 << 64) - 1
  out = bytearray()
  while True:
    b = v & 0x7f
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _ld(field, payload):
  return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
  """features: {name: (BYTES, [bytes...]) | (FLOAT, [floats]) | (INT64, [ints])}.  Entries are
  written in sorted key order (protobuf's deterministic map serialisation)."""
  body = b""
  for name in sorted(features):
    kind, values = features[name]
    if kind == BYTES:
      lst = b"".join(_ld(1, v if isinstance(v, bytes) else v.encode()) for v in values)
    elif kind == FLOAT:
      lst = _ld(1, struct.pack("<%df" % len(values), *values)) if len(values) else b""
    elif kind == INT64:
      lst = _ld(1, b"".join(_varint(int(v)) for v in values)) if len(values) else b""
    else:
      raise ValueError(kind)
    feature = _ld(kind, lst)
    entry = _ld(1, name.encode()) + _ld(2, feature)
    body += _ld(1, entry)
  return _ld(1, body)


def decode_jpeg(data):
  """tf.image.decode_jpeg(data, channels=3): [H, W, 3] uint8 numpy array."""
  lib = _lib.load()
  h, w, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  rc = lib.c2d_jpeg_info(data, len(data), ctypes.byref(h), ctypes.byref(w), ctypes.byref(c))
  _check(rc, "JPEG header")
  out = np.empty((h.value, w.value, 3), np.uint8)
  nb = lib.c2d_jpeg_workspace_bytes(h.value, w.value)
  ws = np.empty(nb, np.uint8)
  rc = lib.c2d_jpeg_decode_rgb(data, len(data), out.ctypes.data, h.value, w.value, ws.ctypes.data,
                               nb)
  _check(rc, "JPEG data")
  return out


def to_hash_bucket(s, num_buckets):
  """tf.strings.to_hash_bucket."""
  b = s if isinstance(s, bytes) else s.encode()
  return int(_lib.load().c2d_tf_hash64(b, len(b)) % num_buckets)
