"""Tensor-dict key names at the model boundary (values as in the reference's
core/standard_fields.py:67-132; only the fields the hot path touches)."""


class TFExampleDataFields(object):
  """Feature names of the tf.Example records (core/standard_fields.py:35-65)."""
  image_id = "image/source_id"
  image_encoded = "image/encoded"
  caption_string = "image/caption/string"
  caption_offset = "image/caption/offset"
  caption_length = "image/caption/length"
  number_of_proposals = "image/proposal/num_proposals"
  proposal_box = "image/proposal/bbox"
  proposal_box_ymin = "image/proposal/bbox/ymin"
  proposal_box_xmin = "image/proposal/bbox/xmin"
  proposal_box_ymax = "image/proposal/bbox/ymax"
  proposal_box_xmax = "image/proposal/bbox/xmax"
  object_box = "image/object/bbox"
  object_text = "image/object/class/text"
  object_label = "image/object/class/label"
  object_box_ymin = "image/object/bbox/ymin"
  object_box_xmin = "image/object/bbox/xmin"
  object_box_ymax = "image/object/bbox/ymax"
  object_box_xmax = "image/object/bbox/xmax"


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
