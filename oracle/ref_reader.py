"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's input pipeline:
readers/cap2det_reader.py:31-267, core/preprocess.py:48-78,151-214, core/builder.py:70-128,
core/box_utils.py:9-42.

Independent of csrc/io_native.cpp on purpose: the TFRecord framing and the tf.Example wire
format are decoded here in pure Python, JPEG decoding goes through Pillow = the system
libjpeg-turbo (the library TensorFlow's decode_jpeg op itself links, default IDCT + fancy
upsampling), so the native decoder is pinned against the real thing.  The record/proto formats
and the TF1 legacy resize are third-party semantics (parity unpinned: no reference test, no
TensorFlow here); CRC-32C is pinned by the RFC 3720 test vectors in tests/test_reader_host.py.
"""
import io
import struct

import numpy as np

from oracle import ref_postprocess


def crc32c(data):
  crc = 0xffffffff
  for b in data:
    crc ^= b
    for _ in range(8):
      crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
  return crc ^ 0xffffffff


def masked_crc(data):
  c = crc32c(data)
  return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def read_records(blob, verify=True):
  pos, out = 0, []
  while pos < len(blob):
    (n,) = struct.unpack_from("<Q", blob, pos)
    (lcrc,) = struct.unpack_from("<I", blob, pos + 8)
    payload = blob[pos + 12:pos + 12 + n]
    (dcrc,) = struct.unpack_from("<I", blob, pos + 12 + n)
    if verify:
      assert masked_crc(blob[pos:pos + 8]) == lcrc and masked_crc(payload) == dcrc
    out.append(payload)
    pos += 16 + n
  return out


def _varint(buf, p):
  v, shift = 0, 0
  while True:
    b = buf[p]; p += 1
    v |= (b & 0x7f) << shift
    if not b & 0x80:
      return v, p
    shift += 7


def _fields(buf):
  p = 0
  while p < len(buf):
    tag, p = _varint(buf, p)
    field, wire = tag >> 3, tag & 7
    if wire == 0:
      v, p = _varint(buf, p)
    elif wire == 2:
      n, p = _varint(buf, p); v = buf[p:p + n]; p += n
    elif wire == 5:
      v = buf[p:p + 4]; p += 4
    elif wire == 1:
      v = buf[p:p + 8]; p += 8
    else:
      raise ValueError(wire)
    yield field, wire, v


def parse_example(record):
  """-> {name: list of bytes | float32 array | int64 array}."""
  out = {}
  for f, w, features in _fields(record):
    if f != 1:
      continue
    for f2, w2, entry in _fields(features):
      if f2 != 1:
        continue
      key, feat = None, b""
      for f3, w3, v in _fields(entry):
        if f3 == 1: key = bytes(v).decode()
        elif f3 == 2: feat = v
      for kind, w4, lst in _fields(feat):
        vals = []
        for f5, w5, v in _fields(lst):
          if kind == 1: vals.append(bytes(v))
          elif kind == 2: vals += list(struct.unpack("<%df" % (len(v) // 4), v))
          elif kind == 3:
            if w5 == 0: vals.append(v)
            else:
              q = 0
              while q < len(v):
                x, q = _varint(v, q); vals.append(x)
        if kind == 2: vals = np.asarray(vals, np.float32)
        if kind == 3: vals = np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in vals], np.int64)
        out[key] = vals
  return out


def decode_jpeg(data):
  from PIL import Image
  return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def boxes_of(parsed, prefix, limit=None):
  cols = [parsed.get(prefix + "/" + k) for k in ("ymin", "xmin", "ymax", "xmax")]
  if any(c is None for c in cols):
    return np.zeros((0, 4), np.float32)
  b = np.stack(cols, -1).astype(np.float32)
  return b if limit is None else b[:limit]


def flip_boxes(b):
  return np.stack([b[:, 0], np.float32(1) - b[:, 3], b[:, 2], np.float32(1) - b[:, 1]], -1)


def process_batch(records, min_dimension, max_num_proposals, flips, batch_scale):
  """records: serialized tf.Examples of one batch; flips: per-example bool; batch_scale: the
  batch_resize_scale_value chosen (or None).  Returns image [B,PH,PW,3] fp32, image_shape,
  proposals [B,N,4], number_of_proposals, object_boxes (padded)."""
  f = np.float32
  imgs, shapes, props, objs = [], [], [], []
  for rec, flip in zip(records, flips):
    p = parse_example(rec)
    img = decode_jpeg(p["image/encoded"][0])
    if flip:
      img = img[:, ::-1]
    h, w = img.shape[:2]
    oh, ow = ref_postprocess.min_dimension_size(h, w, min_dimension)
    imgs.append(ref_postprocess.resize_bilinear_legacy(img.astype(np.float32), oh, ow))
    shapes.append([oh, ow, 3])
    pb = boxes_of(p, "image/proposal/bbox", max_num_proposals)
    ob = boxes_of(p, "image/object/bbox")
    props.append(flip_boxes(pb) if flip else pb)
    objs.append(flip_boxes(ob) if flip else ob)
  shapes = np.asarray(shapes, np.int32)
  ph, pw = shapes[:, 0].max(), shapes[:, 1].max()
  b = len(records)
  canvas = np.zeros((b, ph, pw, 3), np.float32)
  for i, im in enumerate(imgs):
    canvas[i, :im.shape[0], :im.shape[1]] = im
  if batch_scale is not None:
    s = f(batch_scale)
    nh, nw = int(np.round(s * f(ph))), int(np.round(s * f(pw)))
    canvas = np.stack([ref_postprocess.resize_bilinear_legacy(c, nh, nw) for c in canvas])
    shapes = np.stack([[int(np.round(s * f(x[0]))), int(np.round(s * f(x[1]))), 3] for x in shapes]).astype(np.int32)
    ph, pw = nh, nw
  pr = np.zeros((b, max_num_proposals, 4), np.float32)
  mo = max(len(o) for o in objs)
  ob = np.zeros((b, mo, 4), np.float32)
  for i in range(b):
    pr[i, :len(props[i])] = props[i]
    ob[i, :len(objs[i])] = objs[i]
  ih, iw = shapes[:, 0].astype(f)[:, None], shapes[:, 1].astype(f)[:, None]

  def sc(x):
    return np.stack([x[..., 0] * ih / f(ph), x[..., 1] * iw / f(pw), x[..., 2] * ih / f(ph),
                     x[..., 3] * iw / f(pw)], -1).astype(f)

  return dict(image=canvas, image_shape=shapes, proposals=sc(pr),
              number_of_proposals=np.array([len(x) for x in props], np.int32), object_boxes=sc(ob))
