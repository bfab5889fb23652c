"""Benchmark of the Cap2Det hot path: images/s of one full WSOD training step
(forward + losses + backward + Adagrad [+ RCCL all-reduce]) on synthetic 500x500x3 images with
2000 proposals each — BASELINE.json's metric on configs[1] (voc07_groundtruth, Inception-V2,
fp32, B=1 image per GPU).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (implicit-GEMM convolution on
the fp32 MFMA pipe): algorithmic FLOPs of its launches / their duration, timed with HIP events on
the launch stream inside the timed region.  `roofline_roi_crop` is the HBM-bound ROI crop named
by the metric.  `cpu_baseline` times the numpy oracle (a port of the reference semantics — the
TF reference cannot run here) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

IMAGE_HW = 500
NUM_PROPOSALS = 2000
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def synthetic_batch(seed, device, classes, pipeline, num_proposals=NUM_PROPOSALS, batch=1,
                    image_hw=(IMAGE_HW, IMAGE_HW)):
  """SURVEY.md §8d synthetic inputs (seeded): image, proposals, object labels and — for the
  caption-driven configs — a 60-token caption over the extractor's open vocabulary."""
  import numpy as np
  import torch
  from cap2det_amd import synthetic
  rng = np.random.default_rng(seed)
  ex = synthetic.make_examples(rng, batch, image_hw[0], image_hw[1], num_proposals,
                               [num_proposals] * batch, classes)
  vocab = synthetic.caption_vocabulary(pipeline)
  single = [c for c in classes if " " not in c]
  ex["concat_caption_string"] = synthetic.synthetic_captions(
      rng, batch, vocab, tokens=60, must_contain=[single[int(rng.integers(0, len(single)))]])
  out = dict(ex)
  for k in ("image", "proposals", "number_of_proposals"):
    out[k] = torch.from_numpy(ex[k]).to(device).contiguous()
  return out, ex


class KernelTimer(object):
  """Wraps cap2det_amd.hip_ops entry points with torch.cuda events (same stream as the launch)
  and accumulates per-family durations and algorithmic work."""

  def __init__(self):
    import torch
    self.torch = torch
    self.records = []   # (family, work, start_event, end_event, algorithmic HBM bytes)
    self.shapes = []    # (entry point, integer arguments) per record, for --per-call
    self.fused = {}     # family -> launches that also carry a BN/ReLU backward in their epilogue
    self.enabled = False

  def wrap(self, ops):
    t = self

    def conv_work(kind):
      def work(args):
        # conv_fwd(x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw, stride, relu)
        if kind == "fwd":
          n, ih, iw, cin, cout, kh, kw, stride = args[9:17]
          oh, ow = -(-ih // stride), -(-iw // stride)
          return 2.0 * n * oh * ow * cin * cout * kh * kw
        # conv_dgrad(dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride, acc)
        n, ih, iw, cin, cout, kh, kw, stride = args[7:15]
        oh, ow = -(-ih // stride), -(-iw // stride)
        return 2.0 * n * oh * ow * cin * cout * kh * kw
      return work

    def wgrad_work(args):
      # conv_wgrad(x, ldx, xoff, dc, ldc, coff, dw, n, ih, iw, cin, cout, kh, kw, stride)
      n, ih, iw, cin, cout, kh, kw, stride = args[7:15]
      oh, ow = -(-ih // stride), -(-iw // stride)
      return 2.0 * n * oh * ow * cin * cout * kh * kw

    def crop_work(args, kwargs):
      feat, boxes = args[0], args[1]
      crop, pk, ps = args[3:6]
      p = (crop - pk) // ps + 1
      out = kwargs.get("out")
      out_bytes = 2.0 if out is not None and out.dtype == t.torch.bfloat16 else 4.0
      return out_bytes * boxes.shape[0] * p * p * feat.shape[3] + 4.0 * (feat.numel() + boxes.numel())

    def esize(x):
      first = x[0] if isinstance(x, (list, tuple)) else x
      return 2.0 if getattr(first, "dtype", None) == t.torch.bfloat16 else 4.0

    def conv_bytes(args, kind):
      # operand + result bytes of one convolution call, each tensor once (weights included)
      if kind == "fwd":
        n, ih, iw, cin, cout, kh, kw, stride = args[9:17]
      elif kind == "fused_dgrad":
        n, ih, iw, cin, cout, kh, kw, stride = args[12:20]
      else:
        n, ih, iw, cin, cout, kh, kw, stride = args[7:15]
      oh, ow = -(-ih // stride), -(-iw // stride)
      es = esize(args[0])
      act = es * n * (ih * iw * cin + oh * ow * cout)
      w = (4.0 if kind == "wgrad" else es) * kh * kw * cin * cout
      return act + w

    def multi_bytes(rows, cin, couts, es):
      return es * rows * (cin + sum(couts)) + es * cin * sum(couts)

    BYTES = {
        "conv_fwd": lambda a: conv_bytes(a, "fwd"), "conv_dgrad": lambda a: conv_bytes(a, "dgrad"),
        "conv_wgrad": lambda a: conv_bytes(a, "wgrad"), "conv_wgrad_partial": lambda a: conv_bytes(a, "wgrad"),
        "conv_dgrad_bn_relu": lambda a: conv_bytes(a, "fused_dgrad") + esize(a[0]) * a[12] * a[13] * a[14] * a[15],
        "conv1x1_dgrad_multi": lambda a: multi_bytes(a[8], a[9], a[4], esize(a[0])) +
                                         (esize(a[0]) * a[8] * a[9] if a[10] else 0.0),
        "conv1x1_dgrad_multi_bn_relu": lambda a: multi_bytes(a[13], a[14], a[4], esize(a[0])) +
                                                 esize(a[0]) * a[13] * a[14] * (2.0 if a[15] else 1.0),
        "conv1x1_wgrad_multi": lambda a: esize(a[0]) * a[8] * (a[9] + sum(a[7])) + 4.0 * a[9] * sum(a[7]),
        "conv1x1_fwd_multi": lambda a: multi_bytes(a[4], a[5], [o.cout for o in a[3][0]], esize(a[0])),
    }

    def small_rows(name, args):
      """rows of the GEMM <= 16384: the one-tile-per-workgroup domain (conv_gemm.hip run_igemm)"""
      try:
        if name == "conv_fwd":
          n, ih, iw, stride = args[9], args[10], args[11], args[16]
        elif name == "conv_dgrad":
          n, ih, iw, stride = args[7], args[8], args[9], args[14]
          stride = 1                       # (rows of an input gradient = input pixels)
        elif name == "conv_fwd_grouped":
          return True
        elif name == "conv1x1_fwd_multi":
          return args[4] <= 16384
        elif name == "conv1x1_dgrad_multi":
          return args[8] <= 16384
        else:
          return False
        return n * (-(-ih // stride)) * (-(-iw // stride)) <= 16384
      except Exception:
        return False

    def timed(fn, family, work_fn):
      def inner(*args, **kwargs):
        if not t.enabled:
          return fn(*args, **kwargs)
        s = t.torch.cuda.Event(enable_timing=True)
        e = t.torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*args, **kwargs)
        e.record()
        w = work_fn(args, kwargs) if family == "roi_crop_pool_fwd" else work_fn(args)
        # families are kept per operand type: bf16 operands run on the bf16 MFMA kernels
        first = args[0][0] if isinstance(args[0], (list, tuple)) else args[0]
        first_dtype = getattr(first, "dtype", None)
        if fn.__name__ == "conv_fwd_grouped":
          first_dtype = args[0][3]           # (descriptor array, count, flops, storage type)
        low = (family != "roi_crop_pool_fwd" and not family.endswith("_bf16") and
               first_dtype == t.torch.bfloat16)
        try:
          nbytes = BYTES[fn.__name__](args) if fn.__name__ in BYTES else 0.0
        except Exception:
          nbytes = 0.0
        fam = family + ("_bf16" if low else "")
        # fp32 operands on the f32x9 kernels (csrc/igemm_x9.hip): a family of its own, priced against
        # the pipe it runs on (bf16 MFMA / 9), never against the fp32 matrix peak
        if fam == "igemm_nt":
          inst = ops.last_dispatch()
          if inst and all(i.endswith(", 3>") for i in inst):
            fam = "igemm_x9"
        if fam == "wgrad_tn":
          inst = ops.last_dispatch()
          if inst and all(i.startswith(("wgrad1x1_x9_kernel<", "wgrad3x3_x9_kernel<")) for i in inst):
            fam = "wgrad_x9"
        # the bf16 step's single-image first stage runs on its own kernel (igemm_small_kernel<*, 2>:
        # one 32x32 tile per workgroup, launch-bound): its own family, not the ring kernel's
        if fam == "igemm_nt_bf16" and small_rows(fn.__name__, args):
          fam = "igemm_small_bf16"
        t.records.append((fam, w, s, e, nbytes))
        if fn.__name__ in ("conv_dgrad_bn_relu", "conv1x1_dgrad_multi_bn_relu"):
          t.fused[fam] = t.fused.get(fam, 0) + 1
        t.shapes.append((fn.__name__, tuple(a for a in args if isinstance(a, (int, bool)))))
        return r
      return inner

    def multi_work(args):
      # conv1x1_dgrad_multi(dcs, ldcs, coffs, ws, couts, dx, lddx, dxoff, rows, cin, accumulate)
      return 2.0 * args[8] * args[9] * sum(args[4])

    def fused_dgrad_work(args):
      # conv_dgrad_bn_relu(dc, ldc, coff, w, y, ldy, yoff, scale, beta, gamma, dc_out, partials,
      #                    n, ih, iw, cin, cout, kh, kw, stride)
      n, ih, iw, cin, cout, kh, kw, stride = args[12:20]
      oh, ow = -(-ih // stride), -(-iw // stride)
      return 2.0 * n * oh * ow * cin * cout * kh * kw

    def fused_multi_work(args):
      # conv1x1_dgrad_multi_bn_relu(dcs, ldcs, coffs, ws, couts, y, ldy, yoff, prods, dx, lddx,
      #                             dxoff, partials, rows, cin, accumulate)
      return 2.0 * args[13] * args[14] * sum(args[4])

    # (the input-gradient GEMMs with the producer layer's BN/ReLU backward in their epilogue are
    # the same kernels: same family, the GEMM's FLOPs as algorithmic work)
    ops.conv_dgrad_bn_relu = timed(ops.conv_dgrad_bn_relu, "igemm_nt", fused_dgrad_work)
    ops.conv1x1_dgrad_multi_bn_relu = timed(ops.conv1x1_dgrad_multi_bn_relu, "igemm_nt",
                                            fused_multi_work)
    def fwd_multi_work(args):
      # conv1x1_fwd_multi(x, ldx, xoff, outs, rows, cin): outs = (array, count, dtype)
      return 2.0 * args[4] * args[5] * sum(o.cout for o in args[3][0])

    ops.conv1x1_fwd_multi = timed(ops.conv1x1_fwd_multi, "igemm_nt", fwd_multi_work)
    ops.conv1x1_dgrad_multi = timed(ops.conv1x1_dgrad_multi, "igemm_nt", multi_work)
    ops.conv_fwd = timed(ops.conv_fwd, "igemm_nt", conv_work("fwd"))
    ops.conv_fwd_grouped = timed(ops.conv_fwd_grouped, "igemm_nt", lambda args: args[0][2])
    ops.conv_dgrad = timed(ops.conv_dgrad, "igemm_nt", conv_work("dgrad"))
    ops.conv_wgrad = timed(ops.conv_wgrad, "wgrad_tn", wgrad_work)
    # conv1x1_wgrad_multi(x, ldx, xoff, dcs, ldcs, coffs, dws, couts, rows, cin)
    ops.conv1x1_wgrad_multi = timed(ops.conv1x1_wgrad_multi, "wgrad_tn",
                                    lambda a: 2.0 * a[8] * a[9] * sum(a[7]))
    # bf16 mode: split-K slabs (same argument positions as conv_wgrad) + the batched reduction of
    # the slabs, whose time belongs to the filter gradients (it replaces their atomics)
    ops.conv_wgrad_partial = timed(ops.conv_wgrad_partial, "wgrad_tn", wgrad_work)
    reduce_family = ["wgrad_tn"]        # (set to the step's storage mode once it is known)
    t.reduce_family = reduce_family
    inner_reduce = ops.wgrad_reduce_batched
    def reduce_timed(*args, **kwargs):
      return timed(inner_reduce, reduce_family[0], lambda a_: 0.0)(*args, **kwargs)
    ops.wgrad_reduce_batched = reduce_timed
    ops.roi_crop_pool_fwd = timed(ops.roi_crop_pool_fwd, "roi_crop_pool_fwd", crop_work)

  def summary(self):
    out = {}
    for family, work, s, e, nbytes in self.records:
      d = out.setdefault(family, dict(launches=0, work=0.0, ms=0.0, bytes=0.0, ideal_ms=0.0,
                                      hbm_bound_calls=0))
      d["launches"] += 1
      d["work"] += work
      d["ms"] += s.elapsed_time(e)
      d["bytes"] += nbytes
      if not family.startswith("roi_crop") and work > 0:
        # the roofline that binds THIS call: matrix pipe or HBM (operands and result once)
        peak = (PEAK_BF16_MFMA_TFLOPS if family.endswith("_bf16") else
                PEAK_F32X9_TFLOPS if family in ("igemm_x9", "wgrad_x9") else PEAK_FP32_MFMA_TFLOPS)
        t_mfma = work / (peak * 1e12) * 1e3
        t_hbm = nbytes / (PEAK_HBM_GBPS * 1e9) * 1e3
        d["ideal_ms"] += max(t_mfma, t_hbm)
        d["hbm_bound_calls"] += int(t_hbm > t_mfma)
    return out


# fp32 GEMMs as nine bf16 partial products: 2.5 PFLOP/s of bf16 MFMAs / 9 products per fp32 product
PEAK_F32X9_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 9.0


def cpu_baseline(pipeline, classes, num_proposals, budget_s=100.0, warmup=3, max_timed=10):
  """MEASURES full training steps of the CPU restatement of the reference semantics (the TF
  reference cannot run here): oracle/torch_step.train_step — `extract_frcnn_feature` and its
  gradient on torch-CPU (oneDNN convolutions + autograd, fp32, host threads: what
  TensorFlow-CPU's Eigen/MKL-DNN kernels and tf.gradients do for the reference), heads / MIDN /
  OICR / Adagrad in the numpy oracle — SURVEY.md §8d, BASELINE.md §4's protocol: `warmup` (3)
  warm-up + `max_timed` (10) timed full steps at the benchmark's own size (one 500x500 image,
  `num_proposals` proposals), median / p10 / p90, the per-stage split and the CPU ROI-crop rate
  under the same algorithmic-bytes definition.  Bounded: the warm-up shrinks to one step when a
  step is slower than budget_s / 12, and the timed steps stop once `budget_s` of timed work is
  spent (at least one always runs) — the counts actually used are reported."""
  import numpy as np
  import torch
  from oracle import ref_labels, ref_model, torch_step
  from cap2det_amd import synthetic
  from tests import util_model
  # threads: all host cores up to 64 (measured on the 256-thread GPU host: 64 threads 7.5 s/step,
  # 256 threads 120 s/step — oneDNN oversubscribes on the small per-ROI convolutions)
  cores = int(os.environ.get("C2D_CPU_BASELINE_THREADS", "0")) or min(os.cpu_count() or 1, 64)
  torch.set_num_threads(cores)
  rng = np.random.default_rng(0)
  P, d = util_model.oracle_state(0, len(classes), 3, 1.0)
  acc = {k: np.full(v.shape, 0.1, np.float32) for k, v in P.items()}
  ex = synthetic.make_examples(rng, 1, IMAGE_HW, IMAGE_HW, num_proposals, [num_proposals], classes)
  t_lab = time.perf_counter()
  labels = ref_labels.groundtruth_extract(ex["object_texts"], classes)
  t_lab = time.perf_counter() - t_lab
  mask = (rng.uniform(size=(num_proposals, d)) < 0.5).astype(np.uint8)
  opts = ref_model.FrcnnOptions()
  loss_opts = dict(midn_loss_weight=1.0, oicr_loss_weight=0.5, oicr_iterations=3,
                   oicr_iou_threshold=0.6, oicr_use_proba_r_given_c=True)
  mults = [(g.scope, g.multiplier) for g in pipeline.train_config.gradient_multiplier]

  def one(timings=None):
    t0 = time.perf_counter()
    with np.errstate(over="ignore"):
      torch_step.train_step(P, acc, ex, labels, opts, loss_opts, mults, 0.01, 1e-6, mask, timings=timings)
    return time.perf_counter() - t0

  warm = [one()]
  if warm[0] <= budget_s / 12.0:
    warm += [one() for _ in range(warmup - 1)]
  times, stages = [], {}
  for _ in range(max_timed):
    if times and sum(times) + times[-1] > budget_s:
      break
    times.append(one(stages))
  order = sorted(times)
  pick = lambda q: order[min(len(order) - 1, int(q * len(order)))]
  med = order[len(order) // 2]
  split = {k: v / len(times) for k, v in stages.items()}
  split["label_branch(groundtruth match, once)"] = t_lab
  crop_bytes = 4.0 * (num_proposals * 14 * 14 * 576 + 32 * 32 * 576 + num_proposals * 4)
  crop_s = split.get("roi_crop_forward")
  return dict(value=1.0 / med, unit="images/s", cores=int(cores), kind="port",
              step_s={"median": med, "p10": pick(0.1), "p90": pick(0.9), "min": order[0], "max": order[-1],
                      "timed_steps": len(times), "warmup_steps": len(warm), "warmup_step": warm[0]},
              stage_s_per_step=split,
              roi_crop_cpu={"achieved": crop_bytes / crop_s / 1e9 if crop_s else None, "unit": "GB/s",
                            "algorithmic_bytes": crop_bytes,
                            "definition": "unfused crop_and_resize 14x14 (the CPU path materialises it): "
                                          "N*14*14*576*4 written + map + boxes read, SURVEY 8d"},
              sample=("CPU restatement of the reference semantics (not TensorFlow): full fp32 train "
                      "step (fwd + losses + bwd + Adagrad) with the Inception-V2 towers, "
                      "crop_and_resize and pooling on torch-CPU (oneDNN + autograd, %d threads) and "
                      "heads/MIDN/OICR/Adagrad in numpy; 1 image 500x500 with %d proposals; %d warm-up "
                      "+ %d timed full steps, median" % (cores, num_proposals, len(warm), len(times))))


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--config", choices=["c1", "c2", "c3", "c4"], default="c1",
                  help="BASELINE.json configs[k]: c1 voc07_groundtruth fp32 (the headline metric), c2 "
                       "coco17_extend_match bf16, c3 coco17_text_classifier_match fp32, c4 "
                       "flickr30k_text_classifier_match bf16; all with 2000 proposals and the "
                       "caption -> label branch inside the timed step")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-kernel-timing", action="store_true")
  ap.add_argument("--host-inputs", action="store_true",
                  help="secondary measurement (DESIGN.md section 7): every step's image / proposals / counts "
                       "start in pinned HOST memory and are uploaded inside the timed region (the "
                       "headline keeps its inputs resident in HBM)")
  ap.add_argument("--per-call", action="store_true",
                  help="also print one line per timed conv / ROI-crop call (stderr)")
  ap.add_argument("--no-plan", action="store_true",
                  help="A/B: queue every step from Python (~140 ctypes calls) instead of replaying the "
                       "recorded step plan (cap2det_amd/step_plan.py, csrc/plan.hip)")
  ap.add_argument("--image-hw", type=int, nargs=2, default=None, metavar=("H", "W"),
                  help="SECONDARY operating point (never the headline metric): image size, e.g. the "
                       "reference's keep-aspect 1000-px training images (--image-hw 1000 1333)")
  ap.add_argument("--batch", type=int, default=None,
                  help="SECONDARY operating point: images per GPU per step (reference: batch_size 2)")
  ap.add_argument("--proposals", type=int, default=None,
                  help="SECONDARY operating point: proposals per image (reference: max_num_proposals 500)")
  ap.add_argument("--reader", action="store_true",
                  help="SECONDARY measurement (SURVEY.md section 8 row f1): TFRecord shards of synthetic JPEGs + "
                       "proposals written with the repo's own writer -> cap2det_reader (host decode on "
                       "threads, GPU flip / resize / pad) -> Trainer.train; reports reader-fed images/s "
                       "beside the resident-input value of the same process")
  ap.add_argument("--reader-records", type=int, default=64, help="--reader: records per run (cycled)")
  ap.add_argument("--reader-workers", type=int, default=10,
                  help="--reader: map_num_parallel_calls (host decode threads; the shipped configs say 10)")
  ap.add_argument("--available-cus", type=int, default=None,
                  help="CUs the launch plans may count on (c2d_set_available_cus; default 256): what a "
                       "data-parallel rank should say when its RCCL channel kernels hold CUs under the "
                       "backward pass (tools/cu_withhold.py measures the sensitivity)")
  ap.add_argument("--nccl-max-nchannels", type=int, default=None,
                  help="NCCL_MAX_NCHANNELS for the ranks (set before anything touches the GPU): fewer RCCL "
                       "channels = fewer CUs taken from the step")
  ap.add_argument("--nccl-min-nchannels", type=int, default=None, help="NCCL_MIN_NCHANNELS, likewise")
  ap.add_argument("--no-f32x9", action="store_true",
                  help="A/B (fp32 configs): every GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) "
                       "instead of the big forward / input-gradient GEMMs as nine bf16 partial products "
                       "(csrc/igemm_x9.hip; DESIGN.md section 5)")
  ap.add_argument("--dtype", choices=["fp32", "bf16"], default=None,
                  help="override the config's precision: fp32 = exact fp32 everywhere; bf16 = the "
                       "convolution towers behind the stem (first stage, ROI crop output, second stage) "
                       "in bf16 storage / fp32 accumulate")
  return ap.parse_args(argv)


def pin_rank_to_cores(local_rank, local_world):
  """One disjoint, contiguous share of the host cores per local rank (in-process
  os.sched_setaffinity BEFORE torch starts its threads: every thread created later inherits it;
  never an `env` / `taskset` hop in front of a rank, never a re-exec).  Eight ranks that each need
  ~2.2 ms of host time per 3-ms bf16 step, plus RCCL's proxy threads, otherwise wander over the
  same cores.  Returns what was chosen (reported in the JSON line), or None when there is nothing
  to do (one rank, no sched_setaffinity, fewer cores than ranks)."""
  if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
    return None
  if os.environ.get("C2D_NO_AFFINITY") == "1":
    return None
  try:
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
      return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return {"cores_per_rank": per, "rank0_first_core": cores[0], "rank0_last_core": cores[per - 1],
            "host_cores": len(cores), "this_rank": [mine[0], mine[-1]]}
  except OSError:
    return None


def launch_ranks(args, argv):
  """`python bench.py --gpus N` as a plain command: this parent never touches the GPU (it does
  not even import torch: the devices are counted from the KFD topology in sysfs); it starts
  `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (no
  exec), relays its output and returns its exit code.  One process per GPU, as the reference's
  workers (train_wsod.sh:46-88).  With fewer devices than ranks (a 1-GPU box) every rank shares
  cuda:0 over gloo (C2D_BENCH_SAME_DEVICE: code-path validation only, flagged in the JSON)."""
  import socket
  import subprocess
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  env.setdefault("MASTER_ADDR", "127.0.0.1")
  if env.get("C2D_BENCH_STUB") != "1" and env.get("C2D_BENCH_SAME_DEVICE") != "1":
    from cap2det_amd.train.gpu_count import count_visible_gpus     # (sysfs only: no torch, no HIP)
    ndev = count_visible_gpus()
    if ndev == 0:
      print("bench.py: no GPU visible (set C2D_BENCH_STUB=1 for the CPU launcher test)", file=sys.stderr)
      return 2
    if ndev < args.gpus:
      env["C2D_BENCH_SAME_DEVICE"] = "1"
  port = env.get("MASTER_PORT")
  if not port:
    with socket.socket() as sock:
      sock.bind(("127.0.0.1", 0))
      port = str(sock.getsockname()[1])
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
         str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", port,
         os.path.abspath(__file__)] + list(argv)
  proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
  for line in proc.stdout:
    sys.stdout.write(line)
    sys.stdout.flush()
  return proc.wait()


def stub_main(args):
  """C2D_BENCH_STUB=1 (tests/test_bench_launcher.py, CPU): the launcher / rank / barrier /
  max-over-ranks / JSON plumbing of this file with the training step replaced by the
  data-parallel exchange alone — the two-bucket OverlappedReducer over gloo on a CPU bucket of
  the real size — so that the multi-rank path is exercised where no GPU exists."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  affinity = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")),
                               int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
  import torch
  import torch.distributed as dist
  from cap2det_amd.train import data_parallel
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  if os.environ.get("C2D_BENCH_STUB_FAIL") == "1" and rank == world - 1:
    raise SystemExit(3)                      # (launcher test: a failing rank fails the command)
  bucket = torch.zeros(7_100_000)
  split = 1_110_000
  # the per-block cuts of the real bucket, [Mixed_4e | Mixed_5a | 5b | 5c + heads] (Trainer._block_cuts)
  cuts = [0, 1_110_000, 2_030_000, 4_480_000, 7_100_000]
  exposed = []
  def step(i):
    bucket.fill_(float(rank + 1 + i))
    if i % 2 == 0:
      red = data_parallel.OverlappedReducer(bucket, split)
      red.start_tail()
    else:
      # per-block exchange as the backward pass issues it: last block first; every fourth step one
      # block is never announced (finish() must pick it up with the Mixed_4e prefix)
      red = data_parallel.BlockReducer(bucket, cuts)
      for blk in (3, 2, 1):
        if not (i % 4 == 3 and blk == 2):
          red.start(blk)
    t_f = time.perf_counter()
    scale = red.finish()
    exposed.append(time.perf_counter() - t_f)      # (CPU stub: the host clock around finish())
    want = sum(r + 1 + i for r in range(world)) * scale
    got = bucket * scale
    for q in (0, cuts[1], cuts[2], cuts[3] - 1, -1):
      assert abs(float(got[q]) - want) < 1e-6, (i, q, float(got[q]), want)
  for i in range(args.warmup):
    step(i)
  if world > 1:
    dist.barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i)
  if world > 1:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  mine_step = 1000.0 * elapsed / max(args.steps, 1)
  mine_exposed = 1000.0 * sum(exposed[args.warmup:]) / max(args.steps, 1)
  stats = torch.tensor([mine_step, -mine_step, mine_exposed, -mine_exposed], dtype=torch.float64)
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
  if rank == 0:
    print(json.dumps({"metric": "STUB (no training step): launcher + gradient exchange only",
                      "process_group": {"backend": "gloo", "world_size": world,
                                        "step_ms_max_over_ranks": float(stats[0]),
                                        "step_ms_min_over_ranks": -float(stats[1]),
                                        "allreduce_exposed_ms_max_over_ranks": float(stats[2]),
                                        "allreduce_exposed_ms_min_over_ranks": -float(stats[3]),
                                        "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                                        "nccl_min_nchannels": os.environ.get("NCCL_MIN_NCHANNELS")},
                      "available_cus": args.available_cus or 256,
                      "value": world * args.steps / elapsed, "unit": "exchanges/s",
                      "n_gpus": args.gpus, "world_size": world, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "stub": True, "backend": "gloo", "cpu_affinity": affinity,
                      "exchanges": "two-bucket (even steps) and per-block (odd steps) over the real cuts"}))
  if world > 1:
    dist.destroy_process_group()
  return 0


def write_reader_shards(out_dir, rng, classes, vocab, records, image_hw, num_proposals, shards=4):
  """TFRecord shards in the layout of dataset-tools/create_pascal_tf_record.py:147-196 /
  create_coco_tf_record.py:197-242 (the 15 features readers/cap2det_reader.py parses): a structured
  synthetic JPEG (baseline, 4:2:0, quality 90: what PIL writes and the datasets hold), SelectiveSearch-
  shaped proposals, two object texts and a tokenised caption that names one class."""
  import io
  import numpy as np
  from PIL import Image
  from cap2det_amd import synthetic
  from cap2det_amd.readers import tfrecord as T
  h, w = image_hw
  y, x = np.mgrid[0:h, 0:w]
  single = [c for c in classes if " " not in c]
  per_shard = [[] for _ in range(shards)]
  for r in range(records):
    img = np.clip(np.stack([128 + 90 * np.sin(x / (9.0 + r % 7) + c) * np.cos(y / (5.0 + r % 5) - c)
                            for c in range(3)], -1) + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=90)
    pb = synthetic.synthetic_boxes(rng, num_proposals)
    ob = synthetic.synthetic_boxes(rng, 2)
    picks = rng.choice(len(classes), 2, replace=False)
    texts = [classes[i].encode() for i in picks]
    cap = synthetic.synthetic_captions(rng, 1, vocab, tokens=60,
                                       must_contain=[single[int(rng.integers(0, len(single)))]])[0]
    cap = [t for t in cap if t]
    f = {
        "image/height": (T.INT64, [h]), "image/width": (T.INT64, [w]),
        "image/source_id": (T.BYTES, [("%08d" % r).encode()]),
        "image/encoded": (T.BYTES, [buf.getvalue()]), "image/format": (T.BYTES, [b"jpeg"]),
        "image/object/bbox/ymin": (T.FLOAT, ob[:, 0].tolist()), "image/object/bbox/xmin": (T.FLOAT, ob[:, 1].tolist()),
        "image/object/bbox/ymax": (T.FLOAT, ob[:, 2].tolist()), "image/object/bbox/xmax": (T.FLOAT, ob[:, 3].tolist()),
        "image/object/class/text": (T.BYTES, texts), "image/object/class/label": (T.INT64, [int(i) for i in picks]),
        "image/caption/string": (T.BYTES, [t.encode() for t in cap]),
        "image/caption/offset": (T.INT64, [0]), "image/caption/length": (T.INT64, [len(cap)]),
        "image/proposal/bbox/ymin": (T.FLOAT, pb[:, 0].tolist()), "image/proposal/bbox/xmin": (T.FLOAT, pb[:, 1].tolist()),
        "image/proposal/bbox/ymax": (T.FLOAT, pb[:, 2].tolist()), "image/proposal/bbox/xmax": (T.FLOAT, pb[:, 3].tolist()),
    }
    per_shard[r % shards].append(T.encode_example(f))
  paths = []
  for i, recs in enumerate(per_shard):
    paths.append(os.path.join(out_dir, "bench-%05d-of-%05d.record" % (i, shards)))
    T.write_records(paths[-1], recs)
  return paths, sum(len(r) for recs in per_shard for r in recs)


def reader_main(args):
  """`bench.py --reader`: the timed region is Trainer.train over cap2det_reader — record framing +
  CRC, tf.Example parsing and JPEG decoding on host threads, uploads from pinned memory on the input
  thread's copy stream, flip / resize / pad on the GPU, the look-ahead fed from the NEXT batch — at
  the headline shape (or --image-hw / --batch / --proposals).  One JSON line, labelled SECONDARY:
  reader-fed images/s, the resident-input images/s of the same process and their ratio."""
  import shutil
  import tempfile
  import numpy as np
  import torch
  from cap2det_amd import synthetic
  from cap2det_amd.protos import reader_pb2, text_format
  from cap2det_amd.readers import cap2det_reader
  from cap2det_amd.train.trainer import Trainer
  device = "cuda:0"
  torch.cuda.set_device(0)
  spec = synthetic.BASELINE_CONFIGS[args.config]
  if args.dtype is None:
    args.dtype = spec["dtype"]
  image_hw = tuple(args.image_hw) if args.image_hw else (IMAGE_HW, IMAGE_HW)
  images_per_gpu = args.batch or 1
  num_proposals = args.proposals or NUM_PROPOSALS
  scratch = tempfile.mkdtemp(prefix="c2d_bench_reader_")
  try:
    pipeline = synthetic.baseline_pipeline(args.config, scratch)
    trainer = Trainer(pipeline, device=device, seed=1234, compute_dtype=args.dtype,
                      allow_missing_pretrained=True, use_plan=not args.no_plan)
    classes = trainer.model.label_extractor.classes
    rng = np.random.default_rng(77)
    t_w = time.perf_counter()
    paths, nbytes = write_reader_shards(scratch, rng, classes, synthetic.caption_vocabulary(pipeline),
                                        args.reader_records, image_hw, num_proposals)
    t_w = time.perf_counter() - t_w
    opt = reader_pb2.Reader()
    text_format.Merge("""
      cap2det_reader {
        input_pattern: "%s/bench-*.record"
        interleave_cycle_length: 2
        is_training: true
        shuffle_buffer_size: 16
        map_num_parallel_calls: %d
        prefetch_buffer_size: 500
        batch_size: %d
        max_num_proposals: %d
        image_resizer { default_resizer {} }
        preprocess_options { random_flip_left_right_prob: 0.5 }
      }""" % (scratch, args.reader_workers, images_per_gpu, num_proposals), opt)
    input_fn = cap2det_reader.get_input_fn(opt.cap2det_reader, device=device, seed=5)
    total = args.warmup + args.steps
    clock = {}

    def log(step, losses):
      if step == args.warmup:
        torch.cuda.synchronize()
        clock["t0"] = time.perf_counter()

    if args.warmup == 0:             # (no warm-up step to start the clock behind)
      clock["t0"] = time.perf_counter()

    # (a) reader-fed: ONE Trainer.train call; the clock starts behind the warm-up steps
    losses = trainer.train(input_fn(), max_steps=total, log=log)
    torch.cuda.synchronize()
    fed = time.perf_counter() - clock["t0"]
    fed_loss = float(losses["total_loss"].item())
    input_wait, enqueue = trainer.input_wait_s, trainer.enqueue_s
    # (b) the same process on resident inputs: one reader batch kept in HBM, stepped K times with
    # the look-ahead on the same tensors (what the headline command measures)
    it = iter(input_fn())
    batch = next(it)
    it.close()
    torch.cuda.synchronize()
    for i in range(args.warmup):
      trainer.train_step(batch, prefetch=batch if i + 1 < args.warmup else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
      trainer.train_step(batch, prefetch=batch if i + 1 < args.steps else None)
    torch.cuda.synchronize()
    resident = time.perf_counter() - t0
  finally:
    shutil.rmtree(scratch, ignore_errors=True)
  images = args.steps * images_per_gpu
  result = {
      "metric": "SECONDARY (input pipeline, SURVEY section 8 row f1; not the BASELINE metric): images/sec of "
                "Trainer.train fed by cap2det_reader from TFRecord shards (%dx%d JPEG, %d proposals, "
                "%d image(s) per step)" % (image_hw[0], image_hw[1], num_proposals, images_per_gpu),
      "value": images / fed, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1000.0 * fed / args.steps, "higher_is_better": True, "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f32" if args.dtype == "fp32" else "bf16 storage, f32 accumulate",
      "data": "synthetic JPEG records (%d records, %.1f MB, written in %.1f s with cap2det_amd.readers."
              "tfrecord.write_records), decoded inside the timed region" % (args.reader_records, nbytes / 1e6, t_w),
      "config": {"workload": spec["title"] + " fed by the reader", "baseline_config": args.config,
                 "images_per_gpu": images_per_gpu, "image_hw": list(image_hw), "proposals": num_proposals,
                 "reader": "map_num_parallel_calls %d, shuffle 16, flip 0.5, default_resizer, "
                           "input thread + copy stream, 2 batches ahead" % args.reader_workers},
      "resident_inputs": {"value": images / resident, "ms_per_step": 1000.0 * resident / args.steps},
      # host time of the training thread per step, warm-up steps included: waiting for the input
      # thread / queueing the step's launches
      "host_ms_per_step": {"waiting_for_input": 1000.0 * input_wait / total,
                           "queueing_the_step": 1000.0 * enqueue / total},
      "reader_over_resident": resident / fed,
      "steps_replayed_from_a_plan": trainer.plan_replays,
      "final_total_loss": fed_loss,
  }
  print(json.dumps(result))
  sys.stdout.flush()
  return 0


def main(argv=None):
  argv = sys.argv[1:] if argv is None else argv
  args = parse_args(argv)
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(launch_ranks(args, argv))
  if args.nccl_max_nchannels is not None:
    os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_max_nchannels)
  if args.nccl_min_nchannels is not None:
    os.environ["NCCL_MIN_NCHANNELS"] = str(args.nccl_min_nchannels)
  if os.environ.get("C2D_BENCH_STUB") == "1":
    sys.exit(stub_main(args))
  if args.reader:
    if args.gpus != 1:
      raise SystemExit("--reader is a single-GPU secondary measurement")
    sys.exit(reader_main(args))

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  affinity = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
  import torch
  import torch.distributed as dist
  if affinity:
    # (torch sizes its intra-op pool by the host's core count: keep it inside this rank's share)
    torch.set_num_threads(max(1, min(affinity["cores_per_rank"], 16)))
  if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
  # C2D_BENCH_SAME_DEVICE=1 (validation of the multi-rank code path on a 1-GPU box only): every
  # rank uses cuda:0 and the process group runs over gloo; the reported number is meaningless.
  same_device = os.environ.get("C2D_BENCH_SAME_DEVICE") == "1"
  if same_device:
    local_rank = 0
  backend = None
  # C2D_FORCE_ALLREDUCE=1 at one rank: rehearsal of the RCCL path on a 1-GPU box — a process group
  # of one rank over nccl, and the reducers issue their collectives (data_parallel.collectives_on)
  forced = world == 1 and os.environ.get("C2D_FORCE_ALLREDUCE") == "1"
  grouped = world > 1 or forced
  if args.nccl_max_nchannels is not None:
    os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_max_nchannels)
  if args.nccl_min_nchannels is not None:
    os.environ["NCCL_MIN_NCHANNELS"] = str(args.nccl_min_nchannels)
  if grouped:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_PORT" not in os.environ:
      import socket
      with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    torch.cuda.set_device(local_rank)
    backend = "gloo" if same_device else "nccl"
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    assert dist.get_world_size() == world
  device = "cuda:%d" % local_rank
  torch.cuda.set_device(local_rank)

  import shutil
  import tempfile
  from cap2det_amd import hip_ops, synthetic
  from cap2det_amd.train.trainer import Trainer
  from cap2det_amd import _lib as c2d_lib
  from cap2det_amd.train import data_parallel
  if args.available_cus is None and world > 1 and not same_device:
    # ranks of a real multi-GPU run share their CUs with RCCL's channel kernels under the backward
    # pass: plans sized for 224 CUs cost nothing when all 256 are free (10.67-10.82 ms either way)
    # and 14-16 % less than plans sized for 256 when 32 are taken (profiles/r06_cu_withhold.json)
    args.available_cus = 224
  if args.available_cus is not None:
    c2d_lib.call("c2d_set_available_cus", int(args.available_cus))
  if grouped:
    data_parallel.exposed_events = []
  timer = KernelTimer()
  if not args.no_kernel_timing:
    timer.wrap(hip_ops)
  spec = synthetic.BASELINE_CONFIGS[args.config]
  if args.dtype is None:
    args.dtype = spec["dtype"]
  if not args.no_kernel_timing and args.dtype == "bf16":
    timer.reduce_family[0] = "wgrad_tn_bf16"
  scratch = tempfile.mkdtemp(prefix="c2d_bench_")      # synthetic GloVe / classifier files (c3, c4)
  try:
    pipeline = synthetic.baseline_pipeline(args.config, scratch)
    trainer = Trainer(pipeline, device=device, seed=1234, use_plan=not args.no_plan,
                      compute_dtype=args.dtype, allow_missing_pretrained=True)
  finally:
    shutil.rmtree(scratch, ignore_errors=True)
  if args.no_f32x9:
    trainer.model.engine.enable_f32x9(False)
  x9_on = getattr(trainer.model.engine, "_x9", None) is not None
  classes = trainer.model.label_extractor.classes
  assert len(classes) == spec["classes"]
  # the headline workload, unless --image-hw / --batch / --proposals name a secondary point
  image_hw = tuple(args.image_hw) if args.image_hw else (IMAGE_HW, IMAGE_HW)
  images_per_gpu = args.batch or 1
  num_proposals = args.proposals or NUM_PROPOSALS
  secondary = (image_hw, images_per_gpu, num_proposals) != ((IMAGE_HW, IMAGE_HW), 1, NUM_PROPOSALS)
  batch, _ = synthetic_batch(1000 + rank, device, classes, pipeline, num_proposals, images_per_gpu,
                             image_hw)

  def sync():
    if grouped:
      dist.barrier()
    torch.cuda.synchronize()

  # the input pipeline hands the trainer the next batch one step ahead (here: the same synthetic
  # batch), so the frozen first-stage layers of step k+1 run under step k's second stage
  # (the last warm-up step does not look ahead, so every first-stage pass that a timed step
  # consumes is also computed inside the timed region: K passes for K steps)
  for i in range(args.warmup):
    trainer.train_step(batch, dropout_seed=i, prefetch=batch if i + 1 < args.warmup else None)
  sync()
  # Timed region: K steps.  The LAST timed step carries a HIP event pair around every
  # convolution / ROI-crop launch (instrumenting every step costs ~0.5 ms/step of extra gaps), so
  # the roofline numbers come from inside the timed region.
  # one event per step boundary (no host synchronisation): per-step GPU durations for the
  # p10 / p50 / p90 spread (SURVEY.md §8d "timing method")
  second = trainer.model.engine.second
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
  replays_before = trainer.plan_replays
  t0 = time.perf_counter()
  marks[0].record()
  host = None
  if args.host_inputs:
    tensor_keys = [k for k, v in batch.items() if isinstance(v, torch.Tensor)]
    host = {k: batch[k].cpu().pin_memory() for k in tensor_keys}

    copy_stream = torch.cuda.Stream(device=device)

    def upload():
      # on a stream of its own (the copy engine works beside the kernels of the step in flight);
      # the compute stream waits for it — by the time it gets there the copy is long done
      fresh = dict(batch)
      with torch.cuda.stream(copy_stream):
        for k in tensor_keys:
          fresh[k] = host[k].to(device, non_blocking=True)
      torch.cuda.current_stream().wait_stream(copy_stream)
      for k in tensor_keys:
        fresh[k].record_stream(torch.cuda.current_stream())
      return fresh
    cur_batch = upload()
  for i in range(args.steps):
    instrument = (not args.no_kernel_timing) and i == args.steps - 1
    if instrument:
      # the instrumented step runs every kernel on ONE stream so that a kernel's event pair
      # measures that kernel alone (in the other steps the filter gradients overlap the
      # input-gradient GEMMs on a side stream)
      timer.enabled = True
      trainer.use_plan = False
      side, second.side = second.side, None
      alt, second.alt = second.alt, None
      first_net = trainer.model.engine.first
      alt1, first_net.alt = first_net.alt, None
      trainer.model.engine.invalidate_prefetch()          # this step computes its own first stage
    nxt = batch if (not instrument and i + 1 < args.steps and
                    not (i + 2 == args.steps and not args.no_kernel_timing)) else None
    if host is not None:
      # the next step's inputs cross PCIe while this step is being queued (an input pipeline's
      # double buffer); the look-ahead takes the freshly uploaded image
      nxt_up = upload() if i + 1 < args.steps else None
      losses = trainer.train_step(cur_batch, dropout_seed=args.warmup + i,
                                  prefetch=nxt_up if nxt is not None else None)
      cur_batch = nxt_up
    else:
      losses = trainer.train_step(batch, dropout_seed=args.warmup + i, prefetch=nxt)
    if instrument:
      timer.enabled = False
      trainer.use_plan = not args.no_plan
      second.side = side
      second.alt = alt
      first_net.alt = alt1
    marks[i + 1].record()
  host_enqueue = time.perf_counter() - t0       # the host is done queueing; the GPU may still run
  replays_timed = trainer.plan_replays - replays_before
  sync()
  elapsed = time.perf_counter() - t0
  # (the last step carries the per-kernel event pairs and is left out of the spread)
  per_step = sorted(marks[i].elapsed_time(marks[i + 1])
                    for i in range(args.steps - (0 if args.no_kernel_timing else 1)))
  timer.enabled = False
  total_loss = float(losses["total_loss"].item())
  # Host time to ISSUE one step, outside the timed region: with the GPU idle at the start of every
  # step nothing blocks on a full queue, so this is the host's own cost per step (a Python-driven
  # step: ~140 ctypes calls; a planned step: one c2d_plan_replay call) — `host_enqueue_ms_per_step`
  # below is the wall time of the timed loop's queueing, which in a GPU-bound run is the GPU's pace.
  issue = []
  if host is None:
    for i in range(14):
      t_i = time.perf_counter()
      trainer.train_step(batch, dropout_seed=args.warmup + args.steps + i, prefetch=batch)
      issue.append(time.perf_counter() - t_i)
      torch.cuda.synchronize()
    issue = sorted(issue[4:])
  ranks_counted = None
  if grouped:
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # what the process group itself saw: its world size and a rank count summed over it
    ones = torch.ones(1, device=device, dtype=torch.int32)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    ranks_counted = int(ones.item())
    # per rank: its own step time (GPU events, median) and the GPU time its compute stream spent
    # inside the reducers' finish() — between "all gradients queued" and "all collectives done"
    mine_step = per_step[len(per_step) // 2] if per_step else 0.0
    pairs = data_parallel.exposed_events or []
    mine_exposed = (sum(a.elapsed_time(b) for a, b in pairs) / max(len(pairs), 1)) if pairs else 0.0
    stats = torch.tensor([mine_step, -mine_step, mine_exposed, -mine_exposed], device=device,
                         dtype=torch.float64)
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    rank_stats = {"step_ms_max_over_ranks": float(stats[0]), "step_ms_min_over_ranks": -float(stats[1]),
                  "allreduce_exposed_ms_max_over_ranks": float(stats[2]),
                  "allreduce_exposed_ms_min_over_ranks": -float(stats[3])}

  if rank == 0:
    images = world * args.steps * images_per_gpu
    crop_kernel = ("roi_crop_pool2_fwd_rowwalk_kernel (crop_and_resize 14x14 fused with 2x2 max-pool: a "
                   "lane owns a pooled column of a channel quad and walks the crop rows, one horizontal "
                   "lerp per distinct source row)")
    result = {
        "metric": ("images/sec (500x500, 2000 proposals), full WSOD training step" if not secondary else
                   "SECONDARY operating point (not the BASELINE metric): images/sec (%dx%d, %d "
                   "proposals, %d images per GPU), full WSOD training step"
                   % (image_hw[0], image_hw[1], num_proposals, images_per_gpu)),
        "value": images / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        # time the host needed to QUEUE the timed steps (no synchronisation inside): well below
        # ms_per_step = the GPU is the bottleneck, equal to it = the step is launch-bound
        "host_enqueue_ms_per_step": 1000.0 * host_enqueue / args.steps,
        "host_issue_ms_per_step": 1000.0 * issue[len(issue) // 2] if issue else None,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.dtype == "fp32" else "bf16 storage (convolution towers behind the stem: first stage, ROI crop output, second stage), f32 accumulate",
        "data": "synthetic" + (" (inputs uploaded from pinned host memory inside every timed step)"
                               if args.host_inputs else ""),
        "config": {"workload": "%s%s: Inception-V2, %d classes, label extractor %s inside the step, "
                               "OICR x3, Mixed_4e + second stage + heads trainable, %d image(s) %dx%dx3 "
                               "per GPU, %d proposals, %s, Adagrad; fwd+loss+bwd+optimizer%s"
                               % (spec["title"],
                                  "" if args.dtype == spec["dtype"] else " [precision overridden]",
                                  len(classes), type(trainer.model.label_extractor).__name__,
                                  images_per_gpu, image_hw[0], image_hw[1], num_proposals,
                                  "fp32" if args.dtype == "fp32" else
                                  "bf16 storage / fp32 accumulate in both towers behind the fp32 stem (no "
                                  "reduced-precision reference exists: a changed numerical operating point, "
                                  "bounded over 400 steps in profiles/r04_loss_curve.json)",
                                  ("+gloo all-reduce (ranks share cuda:0)" if same_device else "+RCCL all-reduce") if grouped else ""),
                   "baseline_config": args.config, "pipeline": spec["pipeline"] + ".pbtxt",
                   "images_per_gpu": images_per_gpu, "image_hw": list(image_hw),
                   "proposals": num_proposals, "parallelism": "dp%d" % world,
                   "launch": ("step plan replay: %d of the %d timed steps issued by one c2d_plan_replay call "
                              "each, the others (first / last two) queued from Python"
                              % (replays_timed, args.steps)
                              if replays_timed > 0 else "eager (every call queued from Python)"),
                   "gemm_method": ("f32x9: the forward / input-gradient GEMMs of 256 and more 128x128 tiles "
                                   "as nine EXACT bf16 partial products per fp32 product (x = hi + mid + lo, "
                                   "v_mfma_f32_32x32x16_bf16, fp32 accumulate: results within the fp32 "
                                   "kernels' tolerance of the float64 oracle), the stride-1 filter gradients of "
                                   "the second stage likewise; stride-2 filter gradients, first stage and "
                                   "heads on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32)"
                                   if x9_on else
                                   ("fp32 MFMA (v_mfma_f32_32x32x2_f32)" if args.dtype == "fp32"
                                    else "bf16 MFMA (v_mfma_f32_32x32x16_bf16)"))},
        "world_size": world,
        "final_total_loss": total_loss,
    }
    if affinity:
      result["cpu_affinity"] = affinity
    if grouped:
      result["process_group"] = {"backend": backend, "world_size": dist.get_world_size(),
                                 "ranks_counted_by_all_reduce": ranks_counted,
                                 "forced_at_one_rank": forced,
                                 "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                                 "nccl_min_nchannels": os.environ.get("NCCL_MIN_NCHANNELS")}
      result["process_group"].update(rank_stats)
    result["available_cus"] = int(c2d_lib.load().c2d_get_available_cus())
    crop_path = getattr(trainer.model.engine, "last_crop_bwd", None)
    if crop_path:
      result["config"]["roi_crop_backward"] = crop_path
    if per_step:
      pick = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
      result["step_ms_gpu"] = {"p10": pick(0.1), "p50": pick(0.5), "p90": pick(0.9),
                               "steps": len(per_step)}
    # HBM traffic (bytes per kernel launch, averaged over the family) from the committed
    # rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950,
    # + WRITE_SIZE, separate passes; tools/summarize_pmc.py); null when the summary file is
    # absent.  See profiles/README.md.
    traffic, mfma = {}, {}
    low_cfg = "c2" if args.dtype == "bf16" else "c1"
    import glob
    for pattern, into in (("r*_traffic_%s.json" % low_cfg, traffic), ("r*_mfma_%s.json" % low_cfg, mfma)):
      # newest round first; a file that does not parse is REPORTED and the next older one is tried
      # (round 4 committed four truncated summaries and the line silently lost its traffic fields)
      for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
          with open(cand) as f:
            fams = json.load(f)["families"]
          if not fams:
            raise ValueError("no kernel families in the summary")
          into.update(fams)
          result.setdefault("pmc_summaries", []).append(os.path.basename(cand))
          break
        except Exception as e:   # noqa: BLE001 -- reported in the line, never swallowed
          result.setdefault("pmc_summaries_error", []).append(
              "%s: %s: %s" % (os.path.basename(cand), type(e).__name__, e))
      else:
        result.setdefault("pmc_summaries_error", []).append("no profiles/%s parses" % pattern)
    if not args.no_kernel_timing and args.per_call:
      for (family, work, s, e, _nb), (name, ints) in zip(timer.records, timer.shapes):
        ms = s.elapsed_time(e)
        unit = "GB/s" if family.startswith("roi_crop") else "TFLOP/s"
        rate = work / (ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
        print("%-18s %-20s %-52s %8.1f us %8.1f %s" % (family, name, ints, ms * 1e3, rate, unit),
              file=sys.stderr)
    if not args.no_kernel_timing:
      summ = timer.summary()
      low = args.dtype == "bf16"
      def mfma_family(key, kernel, traffic_key, peak):
        f = summ.get(key)
        if not f:
          return None
        tf = f["work"] / (f["ms"] * 1e-3) / 1e12
        return {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                "frac": tf / peak,
                "traffic": traffic.get(traffic_key, {}).get("hbm_bytes_per_launch"),
                "traffic_bytes_per_step": traffic.get(traffic_key, {}).get("hbm_bytes_per_step"),
                # matrix-pipe busy fraction of the family's dispatches from the committed PMC pass
                # (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)): per clock, and of
                # the MFMAs actually issued — padding taps are skipped, `achieved` counts all taps
                "mfma_busy": mfma.get(traffic_key, {}).get("mfma_busy"),
                "calls_per_step": f["launches"], "avg_call_ms": f["ms"] / f["launches"],
                # input-gradient launches whose epilogue also does the producer layer's BN/ReLU
                # backward (mask, scale, per-column sums; the separate bn_relu_bwd launch is gone):
                # their whole duration is in `family_ms_per_step`, only the GEMM's FLOPs in `achieved`
                "calls_with_fused_bn_relu_backward": timer.fused.get(key, 0),
                # per call the binding roofline is max(FLOPs / matrix peak, algorithmic bytes /
                # 8 TB/s): `binding_frac` = sum of those ideal times / measured time; short-K bf16
                # GEMMs (1x1 block-entry convolutions, ~100-300 FLOP/B) are HBM-bound calls
                "binding_frac": (f["ideal_ms"] / f["ms"]) if f["ms"] else None,
                "hbm_bound_calls": f["hbm_bound_calls"],
                "algorithmic_gb_per_step": f["bytes"] / 1e9,
                "family_ms_per_step": f["ms"], "algorithmic_gflop_per_step": f["work"] / 1e9,
                "timed_with": "HIP events around every launch of the last timed step, which runs "
                              "all kernels on one stream (the other steps overlap the filter "
                              "gradients with the input-gradient GEMMs on a side stream)"}

      # (the committed PMC passes are of the default build: with --no-f32x9 on an fp32 network their
      #  "igemm" family holds only the launches f32x9 leaves on the fp32 pipe — no traffic figure then)
      ig32_key = "igemm" if (low or x9_on or "igemm_x9" not in traffic) else "igemm (no PMC pass of this mode)"
      ig32 = mfma_family("igemm_nt", "igemm_nt_kernel<*> + igemm_small_kernel<*> (implicit-GEMM conv "
                         "fwd + dgrad + heads GEMM, fp32 MFMA 32x32x2; a stride-2 dgrad call = 4 launches)",
                         ig32_key, PEAK_FP32_MFMA_TFLOPS)
      wg32 = mfma_family("wgrad_tn", ("wgrad_tn_kernel<*> (what f32x9 leaves on the fp32 pipe: the filter gradients "
                                      "of the single-image first stage and of the heads, launch-latency-bound)"
                                      if x9_on else
                                      "wgrad_tn_kernel<*> + wgrad3x3_kernel<*> (conv filter gradient, fp32 MFMA)"),
                         "wgrad", PEAK_FP32_MFMA_TFLOPS)
      # whole-step view, independent of how kernels overlap: all MFMA work of a step over the
      # median step time (includes every non-GEMM kernel of the step in the denominator)
      gemm_flops = sum(f["work"] for k, f in summ.items() if not k.startswith("roi_crop"))
      if per_step and not low:
        p50 = per_step[len(per_step) // 2] * 1e-3
        result["step_mfma"] = {"algorithmic_gflop_per_step": gemm_flops / 1e9,
                               "achieved": gemm_flops / p50 / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": gemm_flops / p50 / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               "over": "median step time (all kernels of the step); against the fp32 "
                                       "matrix peak whatever pipe the GEMMs ran on (with f32x9 part of "
                                       "the FLOPs run on the bf16 pipe: this is a rate, not a utilisation)"}
      igx9 = mfma_family("igemm_x9", "igemm_ring_kernel<*, 4, *, 3> (f32x9: implicit-GEMM conv fwd + dgrad on "
                         "fp32 operands as nine bf16 partial products, DMA ring; peak = 2.5 PFLOP/s of bf16 "
                         "MFMAs / 9 = the fp32-equivalent rate of the pipe actually used)", "igemm_x9",
                         PEAK_F32X9_TFLOPS)
      if igx9:
        igx9["bound"] = "mfma bf16x9"
      wgx9 = mfma_family("wgrad_x9", "wgrad1x1_x9_kernel<*> + wgrad3x3_x9_kernel<*> (f32x9: filter gradients of the "
                         "second stage's stride-1 convolutions, both fp32 operands split into three bf16 planes "
                         "as they are staged, nine bf16 MFMAs per 16 rows; 3x3: one tap per block over "
                         "pixel-ordered rows, SAME-padding pairs never staged; split-K atomics)", "wgrad_x9",
                         PEAK_F32X9_TFLOPS)
      if wgx9:
        wgx9["bound"] = "mfma bf16x9"
      if not low:
        if wgx9: result["roofline_wgrad_f32x9"] = wgx9
        if igx9:
          # the dominant family; the GEMMs that stay on the fp32 pipe (first stage, heads, launches of
          # fewer than 256 tiles) beside it
          result["roofline"] = igx9
          if ig32: result["roofline_fp32_igemm"] = ig32
        elif ig32:
          result["roofline"] = ig32
        if wg32: result["roofline_wgrad"] = wg32
      else:
        # second stage on bf16 operands (MFMA 32x32x16 bf16, fp32 accumulate); the frozen /
        # Mixed_4e first stage and the heads stay on the fp32 kernels
        ig16 = mfma_family("igemm_nt_bf16", "igemm_ring_kernel<*, 2> (implicit-GEMM conv fwd + dgrad of "
                           "the second stage, bf16 MFMA 32x32x16, fp32 accumulate, direct-to-LDS "
                           "operand stages in a ring of 2-3)", "igemm_bf16",
                           PEAK_BF16_MFMA_TFLOPS)
        wg16 = mfma_family("wgrad_tn_bf16", "wgrad_tn_bf16_kernel<*> + wgrad1x1_bf16_ring_kernel<*> + "
                           "wgrad3x3_bf16_kernel<*> (conv filter gradient, bf16 MFMA 32x32x16 through "
                           "ds_read_b64_tr_b16, split-K atomics)",
                           "wgrad_bf16", PEAK_BF16_MFMA_TFLOPS)
        sm16 = mfma_family("igemm_small_bf16", "igemm_small_kernel<*, 2> + igemm_small_group_kernel<*, 2> + "
                           "short igemm_ring_kernel launches (the single-image first stage behind the "
                           "stem, one 32x32 tile per workgroup with K split over its waves: "
                           "launch-latency-bound)", "igemm_small_bf16", PEAK_BF16_MFMA_TFLOPS)
        if ig16: result["roofline"] = ig16
        if wg16: result["roofline_wgrad"] = wg16
        if sm16: result["roofline_first_stage"] = sm16
        if ig32: result["roofline_fp32_igemm"] = ig32
        if wg32: result["roofline_fp32_wgrad"] = wg32
      rc = summ.get("roi_crop_pool_fwd")
      if rc:
        gbs = rc["work"] / (rc["ms"] * 1e-3) / 1e9
        result["roofline_roi_crop"] = {
            "kernel": crop_kernel,
            "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": gbs / PEAK_HBM_GBPS, "traffic": traffic.get("roi_crop_pool_fwd", {}).get("hbm_bytes_per_launch"),
            "avg_launch_ms": rc["ms"] / rc["launches"],
            "algorithmic_bytes_per_launch": rc["work"] / rc["launches"],
            # (the metric prices it against HBM; the counters say what actually limits it)
            "limiter": "vector ALU and the L1 / address path (64 B/clk/CU: 1.4 GB of tap fetches per "
                       "launch), not HBM: profiles/r03_crop_counters*.json, DESIGN.md section 3"}
    if not args.no_cpu_baseline and world == 1 and not secondary:
      result["cpu_baseline"] = cpu_baseline(pipeline, classes, NUM_PROPOSALS)
      # configs[0] (N = 300, the reference's own CPU-runnable case): the same CPU step, SURVEY §8d
      result["cpu_baseline_c0"] = cpu_baseline(pipeline, classes, 300, budget_s=30.0)
    if world > 1:
      result["backend"] = ("gloo, all ranks on cuda:0 (C2D_BENCH_SAME_DEVICE: code-path validation, "
                           "the value is not a scaling point)" if same_device else "nccl (RCCL)")
    # RCCL writes its version banner to the C stdout of rank 0: flush it first, so that the JSON
    # line is the LAST line this process prints
    try:
      import ctypes
      ctypes.CDLL(None).fflush(None)
    except Exception:
      pass
    print(json.dumps(result))
    sys.stdout.flush()
  if grouped:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
