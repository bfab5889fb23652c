"""Model interface (reference: models/model_base.py:9-74)."""
import abc


class ModelBase(abc.ABC):
  """Model interface; same abstract methods as the reference's ModelBase."""

  def __init__(self, model_proto, is_training=False):
    self._model_proto = model_proto
    self._is_training = is_training

  @abc.abstractmethod
  def build_prediction(self, examples, **kwargs):
    """examples: dict of input tensors keyed by name -> dict of predictions."""

  @abc.abstractmethod
  def build_loss(self, predictions, **kwargs):
    """predictions -> dict of scalar loss tensors keyed by name."""

  @abc.abstractmethod
  def build_evaluation(self, predictions, **kwargs):
    """predictions -> dict of evaluation metrics."""

  def get_variables_to_train(self):
    """Returns the names of all trainable-capable variables (reference: tf.trainable_variables())."""
    return []

  def get_scaffold(self):
    return None
