"""Schema of the reference's `protos/*.proto`, restated field by field (names, numbers,
labels, proto2 defaults, oneofs, extensions) so that unchanged `configs/*.pbtxt` parse.

Source of truth (reference, read-only): protos/pipeline.proto:7-99, protos/model.proto:3-5,
protos/cap2det_model.proto:9-58, protos/frcnn.proto:4-47, protos/hyperparams.proto:7-123,
protos/label_extractor.proto:3-60, protos/post_process.proto:3-15, protos/optimizer.proto:3-41,
protos/reader.proto:6-52, protos/image_resizer.proto, protos/preprocess.proto.
"""
from cap2det_amd.protos.message import FieldDescriptor, Message

_REGISTRY = {}


def get_message_class(name):
  return _REGISTRY[name]


def _msg(name, fields, oneofs=None):
  """fields: list of (name, number, type, label, default[, enum_values])."""
  fds = {}
  member_of = {}
  for oneof, members in (oneofs or {}).items():
    for m in members:
      member_of[m] = oneof
  for spec in fields:
    fname, number, ftype, label, default = spec[:5]
    enum_values = spec[5] if len(spec) > 5 else None
    fds[fname] = FieldDescriptor(fname, number, ftype, label, default, member_of.get(fname),
                                 enum_values, full_name="%s.%s" % (name, fname))
  cls = type(name, (Message,), {
      "_fields": fds, "_oneofs": dict(oneofs or {}), "_extensions": {}, "_name": name})
  for spec in fields:  # expose enum constants like Hyperparams.FC
    if len(spec) > 5 and spec[5]:
      for k, v in spec[5].items():
        setattr(cls, k, v)
  _REGISTRY[name] = cls
  return cls


def _extend(container, owner, number):
  """`extend <container> { optional <owner> ext = number; }` inside message <owner>."""
  fd = FieldDescriptor("ext", number, owner._name, "optional", None, is_extension=True,
                       full_name="%s.ext" % owner._name)
  container._extensions[fd.full_name] = fd
  owner.ext = fd
  return fd


O, R = "optional", "repeated"

# -- model.proto ---------------------------------------------------------------------
Model = _msg("Model", [])

# -- post_process.proto --------------------------------------------------------------
PostProcess = _msg("PostProcess", [
    ("score_thresh", 1, "float", O, 1e-6),
    ("iou_thresh", 2, "float", O, 0.5),
    ("max_size_per_class", 3, "int32", O, 100),
    ("max_total_size", 4, "int32", O, 300),
])

# -- hyperparams.proto ---------------------------------------------------------------
L1Regularizer = _msg("L1Regularizer", [("weight", 1, "float", O, 1.0)])
L2Regularizer = _msg("L2Regularizer", [("weight", 1, "float", O, 1.0)])
Regularizer = _msg("Regularizer", [
    ("l1_regularizer", 1, "L1Regularizer", O, None),
    ("l2_regularizer", 2, "L2Regularizer", O, None),
], {"regularizer_oneof": ["l1_regularizer", "l2_regularizer"]})
TruncatedNormalInitializer = _msg("TruncatedNormalInitializer", [
    ("mean", 1, "float", O, 0.0), ("stddev", 2, "float", O, 1.0)])
VarianceScalingInitializer = _msg("VarianceScalingInitializer", [
    ("factor", 1, "float", O, 2.0), ("uniform", 2, "bool", O, False),
    ("mode", 3, "enum", O, 0, {"FAN_IN": 0, "FAN_OUT": 1, "FAN_AVG": 2})])
RandomNormalInitializer = _msg("RandomNormalInitializer", [
    ("mean", 1, "float", O, 0.0), ("stddev", 2, "float", O, 1.0)])
GlorotNormalInitializer = _msg("GlorotNormalInitializer", [])
GlorotUniformInitializer = _msg("GlorotUniformInitializer", [])
Initializer = _msg("Initializer", [
    ("truncated_normal_initializer", 1, "TruncatedNormalInitializer", O, None),
    ("variance_scaling_initializer", 2, "VarianceScalingInitializer", O, None),
    ("random_normal_initializer", 3, "RandomNormalInitializer", O, None),
    ("glorot_normal_initializer", 4, "GlorotNormalInitializer", O, None),
    ("glorot_uniform_initializer", 5, "GlorotUniformInitializer", O, None),
], {"initializer_oneof": [
    "truncated_normal_initializer", "variance_scaling_initializer",
    "random_normal_initializer", "glorot_normal_initializer", "glorot_uniform_initializer"]})
BatchNorm = _msg("BatchNorm", [
    ("decay", 1, "float", O, 0.999), ("center", 2, "bool", O, True),
    ("scale", 3, "bool", O, False), ("epsilon", 4, "float", O, 0.001),
    ("train", 5, "bool", O, True)])
Hyperparams = _msg("Hyperparams", [
    ("op", 1, "enum", O, 2, {"CONV": 1, "FC": 2}),
    ("regularizer", 2, "Regularizer", O, None),
    ("initializer", 3, "Initializer", O, None),
    ("activation", 4, "enum", O, 1, {"NONE": 0, "RELU": 1, "RELU_6": 2}),
    ("batch_norm", 5, "BatchNorm", O, None),
    ("regularize_depthwise", 6, "bool", O, False),
])

# -- frcnn.proto ---------------------------------------------------------------------
FasterRcnnFeatureExtractor = _msg("FasterRcnnFeatureExtractor", [
    ("type", 1, "string", O, ""),
    ("first_stage_features_stride", 2, "int32", O, 16),
    ("batch_norm_trainable", 3, "bool", O, False),
])
FRCNN = _msg("FRCNN", [
    ("feature_extractor", 1, "FasterRcnnFeatureExtractor", O, None),
    ("inplace_batchnorm_update", 2, "bool", O, False),
    ("initial_crop_size", 3, "int32", O, 0),
    ("maxpool_kernel_size", 4, "int32", O, 0),
    ("maxpool_stride", 5, "int32", O, 0),
    ("dropout_keep_prob", 6, "float", O, 1.0),
    ("dropout_on_feature_map", 7, "bool", O, True),
    ("checkpoint_path", 8, "string", O, ""),
])

# -- label_extractor.proto -----------------------------------------------------------
GroundtruthExtractor = _msg("GroundtruthExtractor", [("label_file", 1, "string", O, "")])
ExactMatchExtractor = _msg("ExactMatchExtractor", [("label_file", 1, "string", O, "")])
ExtendMatchExtractor = _msg("ExtendMatchExtractor", [("label_file", 1, "string", O, "")])
WordVectorMatchExtractor = _msg("WordVectorMatchExtractor", [
    ("label_file", 1, "string", O, ""),
    ("open_vocabulary_file", 2, "string", O, ""),
    ("open_vocabulary_word_embedding_file", 3, "string", O, ""),
])
TextClassifierMatchExtractor = _msg("TextClassifierMatchExtractor", [
    ("label_file", 1, "string", O, ""),
    ("open_vocabulary_file", 2, "string", O, ""),
    ("open_vocabulary_word_embedding_file", 3, "string", O, ""),
    ("text_classifier_checkpoint_file", 4, "string", O, ""),
    ("hidden_units", 5, "int32", O, 300),
    ("dropout_keep_proba", 6, "float", O, 1.0),
    ("regularizer", 8, "float", O, 1e-6),
    ("label_threshold", 7, "float", O, 0.5),
])
LabelExtractor = _msg("LabelExtractor", [
    ("groundtruth_extractor", 1, "GroundtruthExtractor", O, None),
    ("exact_match_extractor", 2, "ExactMatchExtractor", O, None),
    ("extend_match_extractor", 3, "ExtendMatchExtractor", O, None),
    ("word_vector_match_extractor", 4, "WordVectorMatchExtractor", O, None),
    ("text_classifier_match_extractor", 5, "TextClassifierMatchExtractor", O, None),
], {"label_extractor_oneof": [
    "groundtruth_extractor", "exact_match_extractor", "extend_match_extractor",
    "word_vector_match_extractor", "text_classifier_match_extractor"]})

# -- cap2det_model.proto -------------------------------------------------------------
Cap2DetModel = _msg("Cap2DetModel", [
    ("midn_loss_weight", 91, "float", O, 1.0),
    ("oicr_loss_weight", 92, "float", O, 1.0),
    ("frcnn_options", 1, "FRCNN", O, None),
    ("fc_hyperparams", 12, "Hyperparams", O, None),
    ("oicr_iterations", 21, "int32", O, 0),
    ("oicr_iou_threshold", 22, "float", O, 0.5),
    ("midn_post_processor", 31, "PostProcess", O, None),
    ("oicr_post_processor", 32, "PostProcess", O, None),
    ("eval_min_dimension", 34, "int32", R, None),
    ("oicr_use_proba_r_given_c", 36, "bool", O, True),
    ("label_extractor", 93, "LabelExtractor", O, None),
])
_extend(Model, Cap2DetModel, 1454)
TextModel = _msg("TextModel", [
    ("label_extractor", 1, "GroundtruthExtractor", O, None),
    ("text_classifier", 2, "TextClassifierMatchExtractor", O, None),
])
_extend(Model, TextModel, 1453)

# -- optimizer.proto -----------------------------------------------------------------
GradientDescentOptimizer = _msg("GradientDescentOptimizer", [
    ("use_locking", 1, "bool", O, False)])
AdagradOptimizer = _msg("AdagradOptimizer", [
    ("initial_accumulator_value", 1, "float", O, 0.1), ("use_locking", 2, "bool", O, False)])
AdamOptimizer = _msg("AdamOptimizer", [
    ("beta1", 1, "float", O, 0.9), ("beta2", 2, "float", O, 0.999),
    ("epsilon", 3, "float", O, 1e-08), ("use_locking", 4, "bool", O, False)])
RMSPropOptimizer = _msg("RMSPropOptimizer", [
    ("decay", 1, "float", O, 0.9), ("momentum", 2, "float", O, 0.0),
    ("epsilon", 3, "float", O, 1e-10), ("use_locking", 4, "bool", O, False),
    ("centered", 5, "bool", O, False)])
MomentumOptimizer = _msg("MomentumOptimizer", [
    ("momentum", 1, "float", O, 0.0), ("use_locking", 2, "bool", O, False),
    ("use_nesterov", 3, "bool", O, False)])
Optimizer = _msg("Optimizer", [
    ("sgd", 1, "GradientDescentOptimizer", O, None),
    ("adagrad", 2, "AdagradOptimizer", O, None),
    ("adam", 3, "AdamOptimizer", O, None),
    ("rmsprop", 4, "RMSPropOptimizer", O, None),
    ("momentum", 5, "MomentumOptimizer", O, None),
], {"optimizer": ["sgd", "adagrad", "adam", "rmsprop", "momentum"]})

# -- image_resizer.proto / preprocess.proto / reader.proto ---------------------------
DefaultResizer = _msg("DefaultResizer", [])
FixedShapeResizer = _msg("FixedShapeResizer", [
    ("height", 1, "int32", O, 300), ("width", 2, "int32", O, 300)])
KeepAspectRatioResizer = _msg("KeepAspectRatioResizer", [
    ("min_dimension", 3, "int32", O, 600)])
RandomScaleResizer = _msg("RandomScaleResizer", [("max_dimension", 1, "int32", R, None)])
ImageResizer = _msg("ImageResizer", [
    ("default_resizer", 1, "DefaultResizer", O, None),
    ("fixed_shape_resizer", 2, "FixedShapeResizer", O, None),
    ("keep_aspect_ratio_resizer", 3, "KeepAspectRatioResizer", O, None),
    ("random_scale_resizer", 4, "RandomScaleResizer", O, None),
], {"image_resizer_oneof": ["default_resizer", "fixed_shape_resizer",
                            "keep_aspect_ratio_resizer", "random_scale_resizer"]})
Preprocess = _msg("Preprocess", [
    ("random_flip_left_right_prob", 1, "float", O, 0.0),
    ("random_crop_prob", 2, "float", O, 0.0),
    ("random_crop_min_scale", 3, "float", O, 0.8),
    ("random_brightness_prob", 4, "float", O, 0.0),
    ("random_brightness_max_delta", 5, "float", O, 0.2),
    ("random_contrast_prob", 6, "float", O, 0.0),
    ("random_contrast_lower", 7, "float", O, 0.8),
    ("random_contrast_upper", 8, "float", O, 1.2),
    ("random_hue_prob", 9, "float", O, 0.0),
    ("random_hue_max_delta", 10, "float", O, 0.18),
    ("random_saturation_prob", 11, "float", O, 0.0),
    ("random_saturation_lower", 12, "float", O, 0.8),
    ("random_saturation_upper", 13, "float", O, 1.2),
])
Cap2DetReader = _msg("Cap2DetReader", [
    ("input_pattern", 1, "string", R, None),
    ("interleave_cycle_length", 2, "int32", O, 2),
    ("is_training", 3, "bool", O, False),
    ("shuffle_buffer_size", 4, "int32", O, 1000),
    ("map_num_parallel_calls", 5, "int32", O, 1),
    ("prefetch_buffer_size", 6, "int32", O, 200),
    ("batch_size", 7, "int32", O, 32),
    ("decode_image", 11, "bool", O, True),
    ("image_resizer", 12, "ImageResizer", O, None),
    ("preprocess_options", 13, "Preprocess", O, None),
    ("max_num_proposals", 14, "int32", O, 500),
    ("batch_resize_scale_value", 15, "float", R, None),
    ("shard_indicator", 16, "string", O, ""),
])
Reader = _msg("Reader", [("cap2det_reader", 1, "Cap2DetReader", O, None)],
              {"reader_oneof": ["cap2det_reader"]})

# -- pipeline.proto ------------------------------------------------------------------
EvalConfig = _msg("EvalConfig", [
    ("steps", 1, "int32", O, 0),
    ("start_delay_secs", 2, "int32", O, 60),
    ("throttle_secs", 3, "int32", O, 120),
])
LearningRateDecay = _msg("LearningRateDecay", [
    ("decay_steps", 1, "int32", O, 999999999),
    ("decay_rate", 2, "float", O, 1.0),
    ("staircase", 3, "bool", O, True),
])
GradientMultiplier = _msg("GradientMultiplier", [
    ("scope", 1, "string", O, ""), ("multiplier", 2, "float", O, 0.0)])
TrainConfig = _msg("TrainConfig", [
    ("max_steps", 1, "int32", O, 0),
    ("optimizer", 2, "Optimizer", O, None),
    ("learning_rate", 3, "float", O, 0.0),
    ("save_summary_steps", 4, "int32", O, 2000),
    ("save_checkpoints_steps", 5, "int32", O, 2000),
    ("keep_checkpoint_max", 6, "int32", O, 5),
    ("log_step_count_steps", 7, "int32", O, 2000),
    ("learning_rate_decay", 11, "LearningRateDecay", O, None),
    ("sync_replicas", 12, "bool", O, False),
    ("moving_average_decay", 13, "float", O, 0.999),
    ("gradient_multiplier", 16, "GradientMultiplier", R, None),
    ("max_gradient_norm", 17, "float", O, 0.0),
])
Pipeline = _msg("Pipeline", [
    ("train_reader", 1, "Reader", O, None),
    ("eval_reader", 2, "Reader", O, None),
    ("model", 3, "Model", O, None),
    ("model_dir", 4, "string", O, ""),
    ("train_config", 5, "TrainConfig", O, None),
    ("eval_config", 6, "EvalConfig", O, None),
])
