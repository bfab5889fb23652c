"""Tensor-dict key names at the model boundary (values as in the reference's
core/standard_fields.py:67-132; only the fields the hot path touches)."""


class InputDataFields(object):
  """core/standard_fields.py:67-96."""
  image = "image"
  image_id = "image_id"
  image_height = "image_height"
  image_width = "image_width"
  image_shape = "image_shape"
  num_captions = "num_captions"
  caption_strings = "caption_strings"
  caption_lengths = "caption_lengths"
  category_strings = "caption_strings"  # alias kept
  concat_caption_string = "concat_caption_string"
  concat_caption_length = "concat_caption_length"
  num_objects = "number_of_objects"
  object_boxes = "object_boxes"
  object_texts = "object_texts"
  proposals = "proposals"
  num_proposals = "number_of_proposals"


class DetectionResultFields(object):
  """core/standard_fields.py:99-111."""
  num_proposals = "num_proposals"
  proposal_boxes = "proposal_boxes"
  proposal_scores = "proposal_scores"
  class_labels = "class_labels"
  num_detections = "num_detections"
  detection_boxes = "detection_boxes"
  detection_scores = "detection_scores"
  detection_classes = "detection_classes"


class Cap2DetPredictions(object):
  """core/standard_fields.py:124-132."""
  midn_class_logits = "midn_class_logits"
  oicr_proposal_scores = "oicr_proposal_scores"
  midn_proba_r_given_c = "midn_proba_r_given_c"
