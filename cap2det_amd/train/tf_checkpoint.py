"""TensorFlow checkpoint files <-> {variable name: ndarray} ("TF-checkpoint
-> state-dict converter for inception_v2.ckpt and the text checkpoint").

The reference initialises both Inception-V2 towers from `zoo/inception_v2_2016_08_28/
inception_v2.ckpt` (`tf.train.init_from_checkpoint(path, {"/": "<scope>/"})`,
models/utils.py:181-186) and the text classifier from `zoo/coco17_text/model.ckpt-200000`
(models/label_extractor.py:455-457); tf.estimator writes `model.ckpt-<step>` files of the second
kind in `model_dir`.  Two on-disk formats are involved:

  V1 ("inception_v2.ckpt", one file): an SSTable whose values are `SavedTensorSlices` protos;
      key "" holds the `SavedTensorSliceMeta` (names, shapes, types), every other entry one
      `SavedSlice` whose `TensorProto` carries the data in its typed repeated field
      (tensorflow/core/util/tensor_slice_writer.cc, saved_tensor_slice.proto).
  V2 ("model.ckpt-N.index" + "model.ckpt-N.data-0000k-of-0000K"): an SSTable of
      `BundleEntryProto` (dtype, shape, shard, offset, size, masked CRC-32C) over raw
      little-endian tensor bytes; key "" holds the `BundleHeaderProto`
      (tensorflow/core/util/tensor_bundle/, tensor_bundle.proto).

The SSTable is LevelDB's table format (tensorflow/core/lib/io/table*.cc, format.cc): prefix-
compressed key/value blocks with restart arrays, a 5-byte block trailer (compression type +
masked CRC-32C), an index block and a 48-byte footer ending in the magic 0xdb4775248b80fb57;
blocks may be Snappy-compressed.

PARITY UNPINNED: the arithmetic of these formats lives in TensorFlow 1.15 (third party, not
vendored, not installable here) and the reference holds no fixture of either kind, so this
module follows the published formats and is pinned only by its own writer (round trips,
tests/test_tf_checkpoint.py) plus hand-assembled byte vectors for the Snappy / varint / block
layers.  Host code by nature (a one-off conversion, not on the training path).
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_STRING, DT_INT64, DT_BOOL = \
    1, 2, 3, 4, 5, 6, 7, 9, 10
DT_HALF, DT_BFLOAT16 = 19, 14
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8,
             DT_INT16: np.int16, DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_,
             DT_HALF: np.float16}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items()}


class CheckpointError(ValueError):
  """Malformed checkpoint file."""


# ---- CRC-32C (Castagnoli), masked as in tensorflow/core/lib/hash/crc32c.h -----------------------
def _make_crc_table():
  table = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    table.append(c)
  return np.array(table, dtype=np.uint32)


_CRC_TABLE = _make_crc_table()


def crc32c(data):
  """Plain CRC-32C; through the native library when it is loadable (large tensors), otherwise the
  table-driven loop."""
  data = bytes(data)
  try:
    from cap2det_amd import _lib
    return int(_lib.load().c2d_crc32c(data, len(data))) & 0xffffffff
  except Exception:  # pragma: no cover - the library is present wherever the tests run
    c = 0xffffffff
    tab = _CRC_TABLE
    for b in data:
      c = int(tab[(c ^ b) & 0xff]) ^ (c >> 8)
    return c ^ 0xffffffff


def masked_crc32c(data):
  c = crc32c(data)
  return ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


# ---- varints / protobuf wire format ---------------------------------------------------------
def _put_varint(v):
  v &= (1 << 64) - 1
  out = bytearray()
  while v >= 0x80:
    out.append((v & 0x7f) | 0x80)
    v >>= 7
  out.append(v)
  return bytes(out)


def _get_varint(buf, pos):
  v, shift = 0, 0
  while True:
    if pos >= len(buf) or shift > 63:
      raise CheckpointError("bad varint")
    b = buf[pos]
    pos += 1
    v |= (b & 0x7f) << shift
    if not b & 0x80:
      return v, pos
    shift += 7


def _fields(buf):
  """Yields (field number, wire type, value) of one protobuf message; length-delimited values
  come back as bytes, fixed32/64 as ints."""
  pos, n = 0, len(buf)
  while pos < n:
    key, pos = _get_varint(buf, pos)
    f, wt = key >> 3, key & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      if pos + 8 > n:
        raise CheckpointError("truncated fixed64")
      v = struct.unpack_from("<Q", buf, pos)[0]
      pos += 8
    elif wt == 2:
      ln, pos = _get_varint(buf, pos)
      if pos + ln > n:
        raise CheckpointError("truncated field")
      v = bytes(buf[pos:pos + ln])
      pos += ln
    elif wt == 5:
      if pos + 4 > n:
        raise CheckpointError("truncated fixed32")
      v = struct.unpack_from("<I", buf, pos)[0]
      pos += 4
    else:
      raise CheckpointError("unsupported wire type %d" % wt)
    yield f, wt, v


def _tag(field, wt):
  return _put_varint((field << 3) | wt)


def _ld(field, payload):
  return _tag(field, 2) + _put_varint(len(payload)) + payload


def _vi(field, v):
  return _tag(field, 0) + _put_varint(v)


def _signed(v):
  return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
  """TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}."""
  dims = []
  for f, _, v in _fields(buf):
    if f == 2:
      size = 0
      for f2, _, v2 in _fields(v):
        if f2 == 1:
          size = _signed(v2)
      dims.append(size)
  return tuple(dims)


def _encode_shape(shape):
  return b"".join(_ld(2, _vi(1, int(d))) for d in shape)


# ---- Snappy (raw format) ----------------------------------------------------------------------
def snappy_uncompress(buf):
  n, pos = _get_varint(buf, 0)
  out = bytearray()
  end = len(buf)
  while pos < end:
    tag = buf[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = (tag >> 2) + 1
      if ln > 60:
        nb = ln - 60
        if pos + nb > end:
          raise CheckpointError("snappy: truncated literal length")
        ln = int.from_bytes(buf[pos:pos + nb], "little") + 1
        pos += nb
      if pos + ln > end:
        raise CheckpointError("snappy: truncated literal")
      out += buf[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = ((tag >> 2) & 7) + 4
      if pos >= end:
        raise CheckpointError("snappy: truncated copy")
      off = ((tag >> 5) << 8) | buf[pos]
      pos += 1
    elif kind == 2:
      ln = (tag >> 2) + 1
      if pos + 2 > end:
        raise CheckpointError("snappy: truncated copy")
      off = buf[pos] | (buf[pos + 1] << 8)
      pos += 2
    else:
      ln = (tag >> 2) + 1
      if pos + 4 > end:
        raise CheckpointError("snappy: truncated copy")
      off = int.from_bytes(buf[pos:pos + 4], "little")
      pos += 4
    if off == 0 or off > len(out):
      raise CheckpointError("snappy: bad copy offset")
    for _ in range(ln):            # may overlap its own output (run-length style)
      out.append(out[-off])
  if len(out) != n:
    raise CheckpointError("snappy: length mismatch")
  return bytes(out)


# ---- SSTable -----------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
  if offset + size + 5 > len(data):
    raise CheckpointError("table: block past the end of the file")
  body = data[offset:offset + size]
  ctype = data[offset + size]
  if verify:
    want = struct.unpack_from("<I", data, offset + size + 1)[0]
    if masked_crc32c(data[offset:offset + size + 1]) != want:
      raise CheckpointError("table: block checksum mismatch")
  if ctype == 1:
    body = snappy_uncompress(body)
  elif ctype != 0:
    raise CheckpointError("table: unknown block compression %d" % ctype)
  return body


def _block_entries(block):
  if len(block) < 4:
    raise CheckpointError("table: short block")
  nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * nrestart
  if limit < 0:
    raise CheckpointError("table: bad restart array")
  pos, key = 0, b""
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise CheckpointError("table: bad entry")
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    yield key, bytes(block[pos:pos + vlen])
    pos += vlen


def read_table(path, verify=True):
  """All (key, value) pairs of one SSTable file, in key order."""
  with open(path, "rb") as f:
    data = f.read()
  if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
    raise CheckpointError("%s is not a TensorFlow table file (bad magic)" % path)
  foot = data[len(data) - 48:]
  _, p = _get_varint(foot, 0)          # metaindex handle (unused)
  _, p = _get_varint(foot, p)
  ioff, p = _get_varint(foot, p)
  isize, p = _get_varint(foot, p)
  out = []
  for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
    boff, q = _get_varint(handle, 0)
    bsize, q = _get_varint(handle, q)
    out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
  return out


def snappy_compress_literals(data):
  """A valid (if pointless) Snappy stream made of literals only: lets the tests exercise the
  compressed-block path of the reader."""
  out = bytearray(_put_varint(len(data)))
  pos = 0
  while pos < len(data):
    chunk = data[pos:pos + 65536]
    n = len(chunk) - 1
    if n < 60:
      out.append(n << 2)
    elif n < 256:
      out += bytes([60 << 2, n])
    else:
      out += bytes([61 << 2, n & 0xff, n >> 8])
    out += chunk
    pos += len(chunk)
  return bytes(out)


def write_table(path, items, block_size=4096, restart_interval=16, snappy=False):
  """Writes sorted (key, value) pairs as an SSTable (uncompressed blocks, as TensorFlow's
  checkpoint writers produce; snappy=True marks the blocks as Snappy streams)."""
  items = sorted(items)
  out = bytearray()
  index = []

  def flush(entries):
    block, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
      shared = 0
      if i % restart_interval == 0:
        restarts.append(len(block))
      else:
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
          shared += 1
      block += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v))
      block += k[shared:] + v
      prev = k
    if not restarts:
      restarts = [0]
    for r in restarts:
      block += struct.pack("<I", r)
    block += struct.pack("<I", len(restarts))
    off = len(out)
    body = snappy_compress_literals(bytes(block)) if snappy else bytes(block)
    ctype = b"\x01" if snappy else b"\x00"               # kSnappyCompression / kNoCompression
    out.extend(body + ctype)
    out.extend(struct.pack("<I", masked_crc32c(body + ctype)))
    return off, len(body)

  cur, cur_bytes = [], 0
  for k, v in items:
    cur.append((k, v))
    cur_bytes += len(k) + len(v) + 6
    if cur_bytes >= block_size:
      off, size = flush(cur)
      index.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
      cur, cur_bytes = [], 0
  if cur or not index:
    off, size = flush(cur)
    index.append((cur[-1][0] if cur else b"", _put_varint(off) + _put_varint(size)))
  moff, msize = flush([])                                  # empty metaindex block
  ioff, isize = flush(index)
  foot = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
  out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", TABLE_MAGIC))
  with open(path, "wb") as f:
    f.write(bytes(out))


# ---- V2: tensor bundle ---------------------------------------------------------------------------
def _is_v2(path):
  return os.path.exists(path + ".index")


def _read_v2(prefix, verify):
  entries = read_table(prefix + ".index", verify)
  if not entries or entries[0][0] != b"":
    raise CheckpointError("bundle: header entry missing")
  num_shards, endian = 1, 0
  for f, _, v in _fields(entries[0][1]):
    if f == 1:
      num_shards = v
    elif f == 2:
      endian = v
  if endian != 0:
    raise CheckpointError("bundle: big-endian bundles are not supported")
  shards = {}
  out = {}
  for key, val in entries[1:]:
    dtype, shape, shard, offset, size, crc, sliced = 0, (), 0, 0, 0, None, False
    for f, _, v in _fields(val):
      if f == 1:
        dtype = v
      elif f == 2:
        shape = _parse_shape(v)
      elif f == 3:
        shard = v
      elif f == 4:
        offset = v
      elif f == 5:
        size = v
      elif f == 6:
        crc = v
      elif f == 7:
        sliced = True
    if sliced or dtype not in _NP_OF_DT:
      continue           # partitioned variables / string tensors (object graph): not weights
    if shard not in shards:
      name = "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)
      shards[shard] = np.memmap(name, dtype=np.uint8, mode="r")
    raw = shards[shard][offset:offset + size]
    if len(raw) != size:
      raise CheckpointError("bundle: %s lies past the end of its data shard" % key.decode())
    if verify and crc is not None and masked_crc32c(raw.tobytes()) != crc:
      raise CheckpointError("bundle: checksum mismatch for %s" % key.decode())
    npdt = np.dtype(_NP_OF_DT[dtype])
    if int(np.prod(shape, dtype=np.int64)) * npdt.itemsize != size:
      raise CheckpointError("bundle: size of %s does not match its shape" % key.decode())
    out[key.decode()] = np.frombuffer(raw.tobytes(), dtype=npdt).reshape(shape).copy()
  return out


def write_v2(prefix, arrays):
  """Writes {name: ndarray} as a one-shard V2 checkpoint (`prefix.index`, `prefix.data-...`)."""
  items = [(b"", _vi(1, 1) + _vi(2, 0) + _ld(3, _vi(1, 1)))]      # num_shards, LITTLE, version
  offset = 0
  with open(prefix + ".data-00000-of-00001", "wb") as f:
    for name in sorted(arrays):
      a = np.asarray(arrays[name], order="C")        # (ascontiguousarray would make 0-d arrays 1-d)
      if a.dtype not in _DT_OF_NP:
        raise ValueError("unsupported dtype %s for %s" % (a.dtype, name))
      raw = a.tobytes()
      f.write(raw)
      entry = _vi(1, _DT_OF_NP[a.dtype]) + _ld(2, _encode_shape(a.shape))
      if offset:
        entry += _vi(4, offset)
      entry += _vi(5, len(raw)) + _tag(6, 5) + struct.pack("<I", masked_crc32c(raw))
      items.append((name.encode(), entry))
      offset += len(raw)
  write_table(prefix + ".index", items)


# ---- V1: SavedTensorSlices -------------------------------------------------------------------------
def _parse_tensor_proto(buf):
  """TensorProto -> (dtype, flat ndarray)."""
  dtype, content, chunks = 0, None, []
  packed_of = {5: ("<f4", 5), 6: ("<f8", 1), 7: None, 10: None, 11: None, 13: None}
  for f, wt, v in _fields(buf):
    if f == 1:
      dtype = v
    elif f == 4:
      content = v
    elif f in packed_of:
      if f in (5, 6):
        chunks.append(np.frombuffer(v, dtype=packed_of[f][0]) if wt == 2 else
                      np.array([struct.unpack("<f" if f == 5 else "<d",
                                              struct.pack("<I" if f == 5 else "<Q", v))[0]]))
      else:                                   # varint-coded repeated ints
        if wt == 2:
          vals, p = [], 0
          while p < len(v):
            x, p = _get_varint(v, p)
            vals.append(_signed(x))
          chunks.append(np.array(vals, dtype=np.int64))
        else:
          chunks.append(np.array([_signed(v)], dtype=np.int64))
  if dtype not in _NP_OF_DT:
    return dtype, None
  npdt = np.dtype(_NP_OF_DT[dtype])
  if content is not None and len(content):
    return dtype, np.frombuffer(content, dtype=npdt).copy()
  flat = np.concatenate(chunks) if chunks else np.zeros(0)
  return dtype, flat.astype(npdt)


def _parse_slice(buf):
  """TensorSliceProto -> [(start, length or -1)] per dimension."""
  ext = []
  for f, _, v in _fields(buf):
    if f == 1:
      start, length = 0, -1
      for f2, _, v2 in _fields(v):
        if f2 == 1:
          start = _signed(v2)
        elif f2 == 2:
          length = _signed(v2)
      ext.append((start, length))
  return ext


def _read_v1(path, verify):
  entries = read_table(path, verify)
  if not entries or entries[0][0] != b"":
    raise CheckpointError("checkpoint: SavedTensorSliceMeta entry missing")
  shapes, types = {}, {}
  for f, _, v in _fields(entries[0][1]):
    if f != 1:
      continue
    for f2, _, v2 in _fields(v):              # SavedTensorSliceMeta.tensor
      if f2 != 1:
        continue
      name, shape, dtype = None, (), 0
      for f3, _, v3 in _fields(v2):
        if f3 == 1:
          name = v3.decode()
        elif f3 == 2:
          shape = _parse_shape(v3)
        elif f3 == 3:
          dtype = v3
      if name is not None:
        shapes[name], types[name] = shape, dtype
  out = {}
  for key, val in entries[1:]:
    for f, _, v in _fields(val):
      if f != 2:
        continue
      name, ext, flat = None, [], None
      for f2, _, v2 in _fields(v):            # SavedSlice
        if f2 == 1:
          name = v2.decode()
        elif f2 == 2:
          ext = _parse_slice(v2)
        elif f2 == 3:
          _, flat = _parse_tensor_proto(v2)
      if name is None or flat is None or name not in shapes:
        continue
      shape = shapes[name]
      if name not in out:
        out[name] = np.zeros(shape, dtype=flat.dtype)
      idx = tuple(slice(s, None if l < 0 else s + l) for s, l in ext) if ext else ()
      target = out[name][idx] if idx else out[name]
      if target.size != flat.size:
        raise CheckpointError("checkpoint: slice of %s does not match its extent" % name)
      if idx:
        out[name][idx] = flat.reshape(target.shape)
      else:
        out[name][...] = flat.reshape(shape)
  return out


def _ordered_num(v):
  """OrderedCode::WriteNumIncreasing."""
  body = b"" if v == 0 else int(v).to_bytes((int(v).bit_length() + 7) // 8, "big")
  return bytes([len(body)]) + body


def _ordered_signed_small(v):
  """OrderedCode::WriteSignedNumIncreasing for -64 <= v < 64 (one byte)."""
  assert -64 <= v < 64
  return bytes([(0x80 + v) & 0xff])


def _slice_key(name, ndim):
  """EncodeTensorNameSlice(name, full slice) of tensor_slice_writer / saved_tensor_slice_util."""
  esc = name.encode().replace(b"\xff", b"\xff\x00").replace(b"\x00", b"\x00\xff")
  key = _ordered_num(0) + esc + b"\x00\x01" + _ordered_num(ndim)
  for _ in range(ndim):
    key += _ordered_signed_small(0) + _ordered_signed_small(-1)     # start 0, length kFullExtent
  return key


def write_v1(path, arrays):
  """Writes {name: ndarray} as a V1 checkpoint (one full slice per tensor)."""
  meta, items = b"", []
  for name in sorted(arrays):
    a = np.asarray(arrays[name], order="C")        # (ascontiguousarray would make 0-d arrays 1-d)
    if a.dtype not in (np.float32, np.float64, np.int32, np.int64):
      raise ValueError("unsupported dtype %s for %s" % (a.dtype, name))
    dt = _DT_OF_NP[a.dtype]
    full = b"".join(_ld(1, b"") for _ in range(a.ndim))             # Extent{} per dimension
    meta += _ld(1, _ld(1, name.encode()) + _ld(2, _encode_shape(a.shape)) + _vi(3, dt) +
                _ld(4, full))
    tp = _vi(1, dt) + _ld(2, _encode_shape(a.shape))
    if a.dtype == np.float32:
      tp += _ld(5, a.astype("<f4").tobytes())
    elif a.dtype == np.float64:
      tp += _ld(6, a.astype("<f8").tobytes())
    else:
      tp += _ld(7 if a.dtype == np.int32 else 10,
                b"".join(_put_varint(int(x)) for x in a.ravel()))
    items.append((_slice_key(name, a.ndim),
                  _ld(2, _ld(1, name.encode()) + _ld(2, full) + _ld(3, tp))))
  items.append((b"", _ld(1, meta + _ld(2, _vi(1, 0)))))
  write_table(path, items)


# ---- public entry points -------------------------------------------------------------------------------
def read_checkpoint(path, verify_crc=True):
  """{variable name: ndarray} of a V2 prefix (`model.ckpt-200000`) or a V1 file
  (`inception_v2.ckpt`)."""
  if _is_v2(path):
    return _read_v2(path, verify_crc)
  if os.path.isfile(path):
    return _read_v1(path, verify_crc)
  raise FileNotFoundError(path)


def checkpoint_exists(path):
  return bool(path) and (_is_v2(path) or os.path.isfile(path))


def assignment(arrays, variable_names, scope):
  """`tf.train.init_from_checkpoint(path, {"/": scope + "/"})` (models/utils.py:181-186): every
  model variable `scope/X` takes checkpoint tensor `X`; a variable the checkpoint lacks is an
  error, checkpoint tensors the model lacks (logits, Mixed_5 of the first tower, optimizer
  slots) are ignored."""
  out = {}
  prefix = scope.rstrip("/") + "/"
  for name in variable_names:
    if not name.startswith(prefix):
      continue
    src = name[len(prefix):]
    if src not in arrays:
      raise ValueError("Tensor %s is not found in the checkpoint (wanted by %s)" % (src, name))
    out[name] = arrays[src]
  return out
