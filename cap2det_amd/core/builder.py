"""Function builders (reference: core/builder.py:15-67 `build_post_processor`).

`build_post_processor(options)` returns the reference's callable
`(boxes [B,N,4], scores [B,N,C]) -> (num_detections, nmsed_boxes, nmsed_scores,
nmsed_classes (1-based), additional_fields)`; the suppression runs in the HIP kernels behind
`c2d_multiclass_nms` instead of object_detection's `batch_multiclass_non_max_suppression`.
"""
from cap2det_amd import hip_ops as ops
from cap2det_amd.protos import post_process_pb2
from cap2det_amd.protos.message import unwrap


def build_post_processor(options):
  """core/builder.py:15-67."""
  options = unwrap(options)
  if not isinstance(options, post_process_pb2.PostProcess):
    raise ValueError('The options has to be an instance of post_process_pb2.PostProcess.')

  def _post_process(boxes, scores, additional_fields=None, ld=None, off=0, num_classes=None):
    """boxes [B,N,4]; scores [B,N,C] contiguous, or (with ld/off/num_classes) the class columns
    [off, off+num_classes) of a wider [B,N,ld] buffer."""
    if additional_fields is not None:
      raise NotImplementedError('additional_fields are not used by the Cap2Det model')
    if ld is None:
      ld, num_classes = scores.shape[-1], scores.shape[-1]
    num, nb, ns, nc = ops.multiclass_nms(
        boxes.contiguous(), scores, ld, off, num_classes, options.score_thresh, options.iou_thresh,
        options.max_size_per_class, options.max_total_size)
    return num, nb, ns, nc, None

  return _post_process
