"""Model registry (reference: models/registry.py:6-30)."""
import logging

_registry = {}


def register_model_class(cid, cls):
  """Registers a model class under its proto extension descriptor (models/registry.py:11-21)."""
  _registry[cid] = cls
  logging.info('Function registered: %s', getattr(cid, "full_name", cid))


def get_registered_model_classes():
  """Returns the dict mapping class ids to classes (models/registry.py:24-30)."""
  return _registry
