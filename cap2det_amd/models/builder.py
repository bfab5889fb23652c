"""Model factory (reference: models/builder.py:13-37): picks the registered class whose proto
extension is set on the `Model` message."""
from cap2det_amd.protos import model_pb2
from cap2det_amd.protos.message import unwrap
from cap2det_amd.models.registry import get_registered_model_classes

import cap2det_amd.models.cap2det_model  # noqa: F401  (registration side effect, builder.py:9-10)
import cap2det_amd.models.text_model     # noqa: F401


def build(options, is_training=False, **kwargs):
  """Builds a Model based on the options.

  Args:
    options: a model_pb2.Model instance.
    is_training: True if this model is being built for training.
  Raises:
    ValueError: if options is invalid (wrong type, or no registered extension is set).
  """
  options = unwrap(options)
  if not isinstance(options, model_pb2.Model):
    raise ValueError('The options has to be an instance of model_pb2.Model.')
  lookup_table = get_registered_model_classes()
  extension = None
  for extension, value in options.ListFields():
    if extension in lookup_table:
      return lookup_table[extension](value, is_training, **kwargs)
  raise ValueError('Unknown model {}, did you forget to call register_model_class?'.format(
      extension.full_name if extension is not None else '<empty>'))
