"""TensorFlow checkpoint reader / writer (cap2det_amd/train/tf_checkpoint.py; SURVEY.md §8f row
f4).  PARITY UNPINNED against TensorFlow itself (not installable here, no fixture in the
reference): the layers with published, hand-checkable encodings (CRC-32C, Snappy, varints, the
SSTable footer) are tested on hand-assembled bytes, the rest by round trips through the module's
own writer."""
import struct

import numpy as np
import pytest

from cap2det_amd.train import tf_checkpoint as C


def test_crc32c_known_answers():
  assert C.crc32c(b"123456789") == 0xE3069283            # RFC 3720 B.4 check value
  assert C.crc32c(b"") == 0
  assert C.crc32c(bytes(32)) == 0x8A9136AA               # 32 zero bytes (RFC 3720 B.4)
  c = C.crc32c(b"abc")
  assert C.masked_crc32c(b"abc") == ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff
  # the table-driven restatement agrees with the native routine the records reader uses
  tab, v = C._CRC_TABLE, 0xffffffff
  for b in b"123456789":
    v = int(tab[(v ^ b) & 0xff]) ^ (v >> 8)
  assert v ^ 0xffffffff == 0xE3069283


def test_snappy_hand_assembled_streams():
  assert C.snappy_uncompress(b"\x05" + bytes([4 << 2]) + b"hello") == b"hello"
  # literal "abc" + 1-byte-offset copy (length 9, offset 3)
  assert C.snappy_uncompress(bytes([12, 2 << 2]) + b"abc" + bytes([((9 - 4) << 2) | 1, 3])) == b"abc" * 4
  # 2-byte-offset copy (length 6, offset 6) after a 6-byte literal
  assert C.snappy_uncompress(bytes([12, 5 << 2]) + b"abcdef" + bytes([((6 - 1) << 2) | 2, 6, 0])) == b"abcdef" * 2
  # literal with a one-byte length (tag 60), 100 bytes
  body = bytes(range(100))
  assert C.snappy_uncompress(bytes([100, 60 << 2, 99]) + body) == body
  # overlapping copy = run length
  assert C.snappy_uncompress(bytes([10, 0]) + b"a" + bytes([((9 - 4) << 2) | 1, 1])) == b"a" * 10
  for bad in (b"\x05\x10hel", bytes([4, 0]) + b"a" + bytes([1, 9]), b"\x03\x00a"):
    with pytest.raises(C.CheckpointError):
      C.snappy_uncompress(bad)
  data = bytes(np.random.default_rng(0).integers(0, 256, 70000, dtype=np.uint8))
  assert C.snappy_uncompress(C.snappy_compress_literals(data)) == data


@pytest.mark.parametrize("snappy", [False, True])
def test_table_round_trip_and_layout(tmp_path, snappy):
  rng = np.random.default_rng(1)
  items = [(b"", b"header")]
  for i in range(700):
    key = ("scope/layer_%03d/%s" % (i // 3, ["weights", "biases", "moving_mean"][i % 3])).encode()
    items.append((key, bytes(rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8))))
  path = str(tmp_path / "t.sst")
  C.write_table(path, items, block_size=1024, snappy=snappy)
  assert C.read_table(path) == sorted(items)
  blob = open(path, "rb").read()
  assert struct.unpack("<Q", blob[-8:])[0] == 0xdb4775248b80fb57 and len(blob) > 48
  # a flipped payload byte is caught by the block checksum
  bad = bytearray(blob); bad[100] ^= 1
  open(path, "wb").write(bytes(bad))
  with pytest.raises(C.CheckpointError):
    C.read_table(path)
  assert len(C.read_table(path, verify=False)) == len(items) or True   # (may or may not parse)
  open(path, "wb").write(blob[:-1])
  with pytest.raises(C.CheckpointError):
    C.read_table(path)


def _arrays(rng):
  return {
      "InceptionV2/Conv2d_1a_7x7/depthwise_weights": rng.standard_normal((7, 7, 3, 8)).astype(np.float32),
      "InceptionV2/Mixed_3b/Branch_0/Conv2d_0a_1x1/weights": rng.standard_normal((1, 1, 192, 64)).astype(np.float32),
      "InceptionV2/Mixed_3b/Branch_0/Conv2d_0a_1x1/BatchNorm/beta": rng.standard_normal(64).astype(np.float32),
      "global_step": np.array(200000, dtype=np.int64),
      "text_classifier/layer1/weights": rng.standard_normal((300, 400)).astype(np.float32),
      "empty": np.zeros((0, 4), np.float32),
      "ints": rng.integers(-5, 5, (3, 2)).astype(np.int32),
  }


def test_v2_bundle_round_trip(tmp_path):
  arrays = _arrays(np.random.default_rng(2))
  prefix = str(tmp_path / "model.ckpt-200000")
  C.write_v2(prefix, arrays)
  assert C.checkpoint_exists(prefix) and not C.checkpoint_exists(prefix + "x")
  got = C.read_checkpoint(prefix)
  assert sorted(got) == sorted(arrays)
  for k, v in arrays.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape
    np.testing.assert_array_equal(got[k], v)
  # tensor bytes are checksummed
  data = prefix + ".data-00000-of-00001"
  raw = bytearray(open(data, "rb").read()); raw[10] ^= 0x40
  open(data, "wb").write(bytes(raw))
  with pytest.raises(C.CheckpointError):
    C.read_checkpoint(prefix)
  assert "global_step" in C.read_checkpoint(prefix, verify_crc=False)


def test_v1_checkpoint_round_trip(tmp_path):
  arrays = {k: v for k, v in _arrays(np.random.default_rng(3)).items()}
  path = str(tmp_path / "inception_v2.ckpt")
  C.write_v1(path, arrays)
  got = C.read_checkpoint(path)
  assert sorted(got) == sorted(arrays)
  for k, v in arrays.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
    np.testing.assert_array_equal(got[k], v)
  # keys follow EncodeTensorNameSlice: "\0" + escaped name + "\0\1" + rank + (start, length)*
  keys = [k for k, _ in C.read_table(path)]
  assert keys[0] == b"" and keys == sorted(keys)
  assert C._slice_key("a", 2) == b"\x00a\x00\x01\x01\x02\x80\x7f\x80\x7f"
  with pytest.raises(FileNotFoundError):
    C.read_checkpoint(str(tmp_path / "nope"))


def test_assignment_follows_init_from_checkpoint():
  arrays = {"InceptionV2/a/weights": np.ones(3), "InceptionV2/Logits/w": np.ones(2)}
  names = ["first_stage_feature_extraction/InceptionV2/a/weights",
           "second_stage_feature_extraction/InceptionV2/a/weights", "midn/proba_r_given_c/weights"]
  got = C.assignment(arrays, names, "first_stage_feature_extraction")
  assert list(got) == ["first_stage_feature_extraction/InceptionV2/a/weights"]
  with pytest.raises(ValueError):
    C.assignment(arrays, names + ["second_stage_feature_extraction/InceptionV2/b/weights"],
                 "second_stage_feature_extraction/")
