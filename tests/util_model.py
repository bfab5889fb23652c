"""Shared helpers for the end-to-end parity tests (seeded inputs, oracle state)."""
import os

import numpy as np

from oracle import ref_model
from cap2det_amd.synthetic import (load_pipeline, make_examples, synthetic_boxes,  # noqa: F401
                                   synthetic_captions)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_state(seed, num_classes, oicr_iterations, dm=1.0, dtype=np.float32, head_std=0.05):
  rng = np.random.default_rng(seed)
  P = ref_model.init_backbone_params(rng, dm=dm, bn_scale=True, randomize_bn=True, dtype=dtype)
  d = ref_model.spec_out_channels(ref_model.SECOND_STAGE,
                                  ref_model.spec_out_channels(ref_model.FIRST_STAGE, 3, dm), dm)
  P.update(ref_model.init_head_params(rng, d, num_classes, oicr_iterations, stddev=head_std,
                                      dtype=dtype))
  for k in P:
    if k.endswith("/biases"):
      P[k] = (0.1 * rng.standard_normal(P[k].shape)).astype(dtype)
  return P, d
