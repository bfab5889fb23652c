"""Generates tests/golden/first_stage_{500x500,1000x1333}.npz: the FLOAT64 first stage —
`extract_frcnn_feature`'s frozen prefix + Mixed_4e (/root/reference/models/utils.py:127-136: the
object_detection Inception-V2 `extract_proposal_features`, stem 7x7/2 separable convolution at
250^2, Conv2d_2b / 2c at 125^2, Mixed_3b-4e at 63^2 / 32^2 for a 500x500 image) — at depth 1.0 on
the benchmark's image size and on the reference's as-shipped 1000x1333 training shape
(configs/voc07_groundtruth.pbtxt:9-23), through oracle/torch_step.py's torch-CPU float64 ops
(pinned against the numpy oracle by tests/test_oracle_vs_torch.py).

Run in the build container:  python tests/golden/gen_first_stage_fixture.py   (≈1 min)
The fixtures hold EXPECTED OUTPUTS only: per top-level op of the first stage the per-channel L2
norms, the largest magnitude and SAMPLES sampled entries of its output map;
tests/test_gpu_first_stage_fixture.py regenerates the seeded image and variables and checks their
checksums against the ones stored here."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

SIZES = [(500, 500), (1000, 1333)]
DM, SEED, SAMPLES = 1.0, 21, 4096


def inputs(hw):
  """Seeded image [1,H,W,3] (fp32 pixel values 0..255) and backbone variables."""
  from oracle import ref_model
  rng = np.random.default_rng(7000 + hw[0] + hw[1])
  # a smooth image plus noise (a uniform-noise image makes every map a near-constant + noise)
  yy, xx = np.meshgrid(np.linspace(0, 6.0, hw[0]), np.linspace(0, 9.0, hw[1]), indexing="ij")
  base = np.stack([np.sin(yy + 0.3 * c) * np.cos(xx * (1.0 + 0.2 * c)) for c in range(3)], -1)
  image = np.clip(127.5 + 90.0 * base + 25.0 * rng.standard_normal(hw + (3,)), 0, 255)
  image = np.round(image).astype(np.float32)[None]
  P32 = ref_model.init_backbone_params(np.random.default_rng(SEED), dm=DM, bn_scale=True,
                                       randomize_bn=True, dtype=np.float32)
  return image, P32


def checksum(image, P32):
  return np.array([float(image.astype(np.float64).sum()),
                   float(sum(np.abs(v.astype(np.float64)).sum() for k, v in sorted(P32.items())
                             if k.startswith("first_stage")))])


def sample_indices(name, size):
  seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 32)
  return np.random.default_rng(seed).integers(0, size, SAMPLES)


def main():
  import torch
  from oracle import ref_model, torch_step
  torch.set_num_threads(8)
  for hw in SIZES:
    image, P32 = inputs(hw)
    T = {k: torch.from_numpy(v.astype(np.float64)) for k, v in P32.items()
         if k.startswith("first_stage")}
    x = torch_step._nchw(image.astype(np.float64)) * (2.0 / 255.0) - 1.0
    arrays = {"checksum": checksum(image, P32)}
    names = []
    with torch.no_grad():
      for op in ref_model.FIRST_STAGE:
        x = torch_step._op(op, x, T, ref_model.FIRST_SCOPE)
        y = x.permute(0, 2, 3, 1).contiguous().numpy()[0]       # [H][W][C]
        name = op[1]
        names.append(name)
        flat = y.reshape(-1)
        arrays[name + "/shape"] = np.array(y.shape, np.int64)
        arrays[name + "/channel_norm"] = np.sqrt((y.reshape(-1, y.shape[-1]) ** 2).sum(0))
        arrays[name + "/absmax"] = np.float64(np.abs(flat).max())
        arrays[name + "/samples"] = flat[sample_indices(name, flat.size)]
        print(hw, name, y.shape, "absmax %.4g" % np.abs(flat).max(),
              "mean |y| %.4g" % np.abs(flat).mean(), "zeros %.3f" % (flat == 0).mean())
    arrays["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "first_stage_%dx%d.npz" % hw)
    np.savez_compressed(path, **arrays)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
  main()
