"""Cap2Det input pipeline (reference: readers/cap2det_reader.py:19-267 `get_input_fn`).

Same stages as the reference's tf.data graph, with the work split for MI355X:
  host threads : TFRecord framing + CRC, tf.Example parsing, JPEG entropy decode -> RGB u8
                 (csrc/io_native.cpp; ctypes releases the GIL, `map_num_parallel_calls` workers)
  GPU          : flip, keep-aspect legacy-bilinear resize, zero padding of the batch canvas,
                 random batch rescale (csrc/preprocess.hip, c2d_resize_bilinear)
  host (tiny)  : box flip / rescale, caption padding, shard filter (string hash).
Random decisions (flip, batch scale, shuffling) come from one seeded numpy Generator — TF's own
random streams cannot be reproduced; the decisions taken are reported under the private keys
'_flip_left_right' and '_batch_scale' so that tests can replay a batch through the oracle.
"""
import glob
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from cap2det_amd import hip_ops as ops
from cap2det_amd.core.standard_fields import InputDataFields, TFExampleDataFields
from cap2det_amd.protos import reader_pb2
from cap2det_amd.protos.message import unwrap
from cap2det_amd.readers import tfrecord

_IMAGE_CHANNELS = 3

_F = TFExampleDataFields
_KEYS = [_F.image_id, _F.image_encoded, _F.caption_string, _F.caption_offset, _F.caption_length,
         _F.object_box_ymin, _F.object_box_xmin, _F.object_box_ymax, _F.object_box_xmax,
         _F.object_label, _F.object_text, _F.proposal_box_ymin, _F.proposal_box_xmin,
         _F.proposal_box_ymax, _F.proposal_box_xmax]


def parse_texts(tokens, offsets, lengths):
  """core/preprocess.py:151-214: texts cut out of the token buffer, padded with "" to the
  longest one."""
  if len(offsets) != len(lengths):
    raise ValueError("Not equal: num_offsets and num_lengths")
  max_len = max([int(l) for l in lengths] + [0])
  strings = []
  for off, length in zip(offsets, lengths):
    text = list(tokens[int(off):int(off) + int(length)])
    strings.append(text + [""] * (max_len - len(text)))
  return len(offsets), strings, np.asarray(lengths, np.int64)


def flip_boxes_left_right(box):
  """core/box_utils.py:29-42."""
  box = np.asarray(box, np.float32).reshape(-1, 4)
  return np.stack([box[:, 0], np.float32(1.0) - box[:, 3], box[:, 2],
                   np.float32(1.0) - box[:, 1]], axis=-1)


def _boxes(parsed, prefix):
  cols = [parsed[prefix + "/" + k] for k in ("ymin", "xmin", "ymax", "xmax")]
  if any(c is None for c in cols):
    return np.zeros((0, 4), np.float32)
  return np.stack(cols, axis=-1).astype(np.float32)


def _rint(x):
  return int(np.round(np.float32(x)))   # tf.round: half to even


def resized_shape(options, height, width):
  """core/builder.py:70-128 image resizers -> (new_height, new_width)."""
  which = options.WhichOneof('image_resizer_oneof')
  if which == 'default_resizer':
    return height, width
  if which == 'fixed_shape_resizer':
    return options.fixed_shape_resizer.height, options.fixed_shape_resizer.width
  if which == 'keep_aspect_ratio_resizer':
    scale = np.float32(options.keep_aspect_ratio_resizer.min_dimension) / np.float32(
        min(height, width))
    return _rint(np.float32(height) * scale), _rint(np.float32(width) * scale)
  raise ValueError('Invalid resizer: {}.'.format(which))


def get_input_fn(options, device="cuda:0", seed=0):
  """Returns a callable that yields batches (dicts keyed by InputDataFields names).  A batch is made
  in two halves — `_batch_host` (numpy / Python: padded_batch, box scaling, every random decision)
  and `_batch_device` (uploads through the pinned staging ring + flip / resize / pad kernels on the
  CURRENT stream) — so that a caller can run the whole generator on an input thread under a copy
  stream (readers/prefetch.py).  (Round 5 also built the host half as a forked child process with a
  shared-memory pixel ring: bitwise the same batches, but 0.56-0.66 of the resident-input rate
  against 0.85-0.99 for the thread form, profiles/r05_experiments/README.md; not in the tree.)"""
  options = unwrap(options)
  if not isinstance(options, reader_pb2.Cap2DetReader):
    raise ValueError('options has to be an instance of Reader.')

  on_gpu = torch.device(device).type == "cuda"
  # Uploads go through a RING of pinned host buffers owned by this input function: a copy from
  # pageable memory is staged by the runtime and holds the calling thread until it is done, a
  # pinned one is a DMA the copy engine runs beside the kernels of the step in flight — and a
  # fresh `pin_memory()` per array is a hipHostMalloc per array (1.2 ms each, tools/bench_reader.py:
  # 3.8 of the reader's 5.8 ms per batch).  A slot is reused once the event behind its last copy
  # has completed (slots = four batches' worth: the wait never happens in practice).
  staging = dict(slots=[dict(buf=None, event=None)
                        for _ in range(4 * (max(int(options.batch_size), 1) + 2))], i=0)

  def _upload(array):
    """numpy -> device on the CURRENT stream."""
    src = torch.from_numpy(np.ascontiguousarray(array))
    if not on_gpu:
      return src.to(device)
    slot = staging["slots"][staging["i"] % len(staging["slots"])]
    staging["i"] += 1
    if slot["event"] is not None:
      slot["event"].synchronize()
    nbytes = src.numel() * src.element_size()
    if slot["buf"] is None or slot["buf"].numel() < nbytes:
      slot["buf"] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
      slot["np"] = slot["buf"].numpy()
    dst = slot["buf"][:nbytes].view(src.dtype).view(src.shape)
    # (a plain memcpy that drops the GIL: torch's CPU copy_ fans 750 KB out over the intra-op thread
    #  pool, 1.8 ms per call beside ten decoding threads)
    np.copyto(slot["np"][:nbytes], np.ascontiguousarray(array).reshape(-1).view(np.uint8))
    out = dst.to(device, non_blocking=True)
    slot["event"] = torch.cuda.Event()
    slot["event"].record()
    return out

  def _parse_host(record, rng_flip):
    """Host part of `_parse_fn` (readers/cap2det_reader.py:31-139)."""
    parsed = tfrecord.parse_example(record, _KEYS)
    if parsed[_F.image_id] is None or (options.decode_image and parsed[_F.image_encoded] is None):
      raise tfrecord.DataError("missing required feature image/source_id / image/encoded")
    tokens = [t.decode("utf8") for t in (parsed[_F.caption_string] or [])]
    offsets = parsed[_F.caption_offset] if parsed[_F.caption_offset] is not None else []
    lengths = parsed[_F.caption_length] if parsed[_F.caption_length] is not None else []
    num_captions, caption_strings, caption_lengths = parse_texts(tokens, offsets, lengths)
    ex = {
        InputDataFields.image_id: parsed[_F.image_id][0].decode("utf8"),
        InputDataFields.num_captions: num_captions,
        InputDataFields.caption_strings: caption_strings,
        InputDataFields.caption_lengths: caption_lengths,
        InputDataFields.concat_caption_string: tokens,
        InputDataFields.concat_caption_length: len(tokens),
    }
    flip = False
    if options.decode_image:
      ex["_image_u8"] = tfrecord.decode_jpeg(parsed[_F.image_encoded][0])
      if options.HasField("preprocess_options"):
        flip = bool(rng_flip < options.preprocess_options.random_flip_left_right_prob)
      h, w = ex["_image_u8"].shape[:2]
      ex[InputDataFields.image_height], ex[InputDataFields.image_width] = h, w
      oh, ow = resized_shape(options.image_resizer, h, w)
      ex[InputDataFields.image_shape] = np.array([oh, ow, _IMAGE_CHANNELS], np.int32)
    ex["_flip_left_right"] = flip
    proposals = _boxes(parsed, _F.proposal_box)[:options.max_num_proposals]
    object_boxes = _boxes(parsed, _F.object_box)
    if flip:
      proposals = flip_boxes_left_right(proposals)
      object_boxes = flip_boxes_left_right(object_boxes)
    ex[InputDataFields.num_proposals] = proposals.shape[0]
    ex[InputDataFields.proposals] = proposals
    ex[InputDataFields.num_objects] = object_boxes.shape[0]
    ex[InputDataFields.object_boxes] = object_boxes
    ex[InputDataFields.object_texts] = [t.decode("utf8") for t in (parsed[_F.object_text] or [])]
    return ex

  def _records(rng):
    files = sorted(f for pattern in options.input_pattern for f in glob.glob(pattern))
    if not files:
      raise ValueError("no input files match %s" % list(options.input_pattern))
    while True:
      order = list(files)
      if options.is_training:
        rng.shuffle(order)
      cycle = max(int(options.interleave_cycle_length), 1)
      pending = list(order)
      active = [tfrecord.iterate_records(pending.pop(0)) for _ in range(min(cycle, len(pending)))]
      while active:                      # files.interleave(TFRecordDataset, cycle_length)
        for it in list(active):
          try:
            yield next(it)
          except StopIteration:
            i = active.index(it)
            if pending:
              active[i] = tfrecord.iterate_records(pending.pop(0))
            else:
              active.pop(i)
      if not options.is_training:
        return

  def _shuffled(records, rng):
    if not options.is_training:
      yield from records
      return
    buf = []
    for r in records:                    # dataset.repeat().shuffle(shuffle_buffer_size)
      if len(buf) < max(int(options.shuffle_buffer_size), 1):
        buf.append(r)
        continue
      i = int(rng.integers(len(buf)))
      yield buf[i]
      buf[i] = r
    rng.shuffle(buf)
    yield from buf

  def _keep(ex):
    """`_filter_fn`, readers/cap2det_reader.py:201-211."""
    if not options.shard_indicator:
      return True
    numer, denom = options.shard_indicator.split('/')
    assert numer.isdigit() and denom.isdigit()
    numer, denom = int(numer), int(denom)
    assert 0 <= numer < denom
    return tfrecord.to_hash_bucket(ex[InputDataFields.image_id], denom) == numer

  def _batch_host(exs, rng):
    """padded_batch + `_batch_scale_box_fn` and every decision of `_batch_resize_image_fn`: the
    host half of a batch (numpy / Python only; the decoded images ride along under '_u8')."""
    b = len(exs)
    n = options.max_num_proposals
    out = {}
    for key in (InputDataFields.image_id, InputDataFields.concat_caption_length):
      out[key] = [e[key] for e in exs]
    for key in (InputDataFields.num_captions, InputDataFields.num_proposals,
                InputDataFields.num_objects):
      out[key] = np.array([e[key] for e in exs], np.int32)
    max_caps = max(e[InputDataFields.num_captions] for e in exs)
    max_len = max([len(s) for e in exs for s in e[InputDataFields.caption_strings]] + [0])
    out[InputDataFields.caption_strings] = [
        [list(s) + [""] * (max_len - len(s)) for s in e[InputDataFields.caption_strings]] +
        [[""] * max_len] * (max_caps - e[InputDataFields.num_captions]) for e in exs]
    cl = np.zeros((b, max_caps), np.int64)
    for i, e in enumerate(exs):
      cl[i, :e[InputDataFields.num_captions]] = e[InputDataFields.caption_lengths]
    out[InputDataFields.caption_lengths] = cl
    max_tok = max(len(e[InputDataFields.concat_caption_string]) for e in exs)
    out[InputDataFields.concat_caption_string] = [
        list(e[InputDataFields.concat_caption_string]) +
        [""] * (max_tok - len(e[InputDataFields.concat_caption_string])) for e in exs]
    max_obj = max(e[InputDataFields.num_objects] for e in exs)
    ob = np.zeros((b, max_obj, 4), np.float32)
    pr = np.zeros((b, n, 4), np.float32)
    for i, e in enumerate(exs):
      ob[i, :e[InputDataFields.num_objects]] = e[InputDataFields.object_boxes]
      pr[i, :e[InputDataFields.num_proposals]] = e[InputDataFields.proposals]
    out[InputDataFields.object_texts] = [
        list(e[InputDataFields.object_texts]) + [""] * (max_obj - e[InputDataFields.num_objects])
        for e in exs]
    out["_flip_left_right"] = [e["_flip_left_right"] for e in exs]
    out["_batch_scale"] = None
    if options.decode_image:
      shapes = np.stack([e[InputDataFields.image_shape] for e in exs])
      ph, pw = int(shapes[:, 0].max()), int(shapes[:, 1].max())
      out["_u8"] = [e["_image_u8"] for e in exs]
      out["_canvas"] = (ph, pw, shapes.copy())          # canvas size + per-image resized shapes
      out["_rescale"] = None
      out[InputDataFields.image_height] = np.array([e[InputDataFields.image_height] for e in exs], np.int32)
      out[InputDataFields.image_width] = np.array([e[InputDataFields.image_width] for e in exs], np.int32)
      if len(options.batch_resize_scale_value) > 0:
        index = int(rng.integers(len(options.batch_resize_scale_value)))
        scale = np.float32(options.batch_resize_scale_value[index])
        out["_batch_scale"] = float(scale)
        nh, nw = _rint(scale * np.float32(ph)), _rint(scale * np.float32(pw))
        out["_rescale"] = (nh, nw)
        ph, pw = nh, nw
        shapes = np.stack([[_rint(scale * np.float32(s[0])), _rint(scale * np.float32(s[1])), s[2]]
                           for s in shapes]).astype(np.int32)
      out[InputDataFields.image_shape] = shapes
      # `_batch_scale_box_fn`: box * img / pad in fp32, in this order
      img_h = shapes[:, 0].astype(np.float32)[:, None]
      img_w = shapes[:, 1].astype(np.float32)[:, None]

      def scale_box(box):
        return np.stack([box[..., 0] * img_h / np.float32(ph), box[..., 1] * img_w / np.float32(pw),
                         box[..., 2] * img_h / np.float32(ph), box[..., 3] * img_w / np.float32(pw)],
                        axis=-1).astype(np.float32)

      ob, pr = scale_box(ob), scale_box(pr)
    out[InputDataFields.object_boxes] = ob
    out[InputDataFields.proposals] = pr
    return out

  def _batch_device(out):
    """The device half of a batch, on the CURRENT stream: uploads, flip + legacy-bilinear resize +
    zero padding into the batch canvas, `_batch_resize_image_fn`'s rescale."""
    images = out.pop("_u8", None)
    if images is not None:
      ph, pw, shapes = out.pop("_canvas")
      rescale = out.pop("_rescale")
      b = len(images)
      canvas = torch.empty(b, ph, pw, _IMAGE_CHANNELS, device=device)
      for i, img in enumerate(images):
        ops.image_resize_pad_u8(_upload(img), out["_flip_left_right"][i], canvas[i], int(shapes[i, 0]),
                                int(shapes[i, 1]))
      if rescale is not None:
        nh, nw = rescale
        resized = torch.empty(b, nh, nw, _IMAGE_CHANNELS, device=device)
        for i in range(b):
          ops.resize_bilinear(canvas[i], nh, nw, out=resized[i])
        canvas = resized
      out[InputDataFields.image] = canvas
    out[InputDataFields.proposals] = _upload(out[InputDataFields.proposals])
    out[InputDataFields.num_proposals] = _upload(out[InputDataFields.num_proposals])
    return out

  def _host_batches():
    rng = np.random.default_rng(seed)
    workers = max(int(options.map_num_parallel_calls), 1)
    batch_size = int(options.batch_size)
    with ThreadPoolExecutor(max_workers=workers) as pool:
      window, exs = [], []
      for record in _shuffled(_records(rng), rng):
        window.append(pool.submit(_parse_host, record, float(rng.uniform())))
        if len(window) < 2 * workers:
          continue
        ex = window.pop(0).result()
        if _keep(ex):
          exs.append(ex)
        if len(exs) == batch_size:
          yield _batch_host(exs, rng)
          exs = []
      for fut in window:
        ex = fut.result()
        if _keep(ex):
          exs.append(ex)
        if len(exs) == batch_size:       # drop_remainder=True
          yield _batch_host(exs, rng)
          exs = []

  def _input_fn():
    for host in _host_batches():
      yield _batch_device(host)

  return _input_fn
