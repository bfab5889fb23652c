import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
  return GOLDEN


def pytest_collection_modifyitems(config, items):
  """GPU tests are skipped (not failed) when no device is visible and -m gpu was not asked."""
  try:
    import torch
    has_gpu = torch.cuda.is_available()
  except Exception:  # pragma: no cover
    has_gpu = False
  if has_gpu:
    return
  skip = pytest.mark.skip(reason="no GPU visible")
  for item in items:
    if "gpu" in item.keywords:
      item.add_marker(skip)
