"""Data-parallel plumbing: one process per GPU, images sharded across ranks, ONE all-reduce of
the flat gradient bucket per step (RCCL over xGMI on MI355X; gloo in the CPU tests).

The reference shards records across its async workers with a hash filter
(`shard_indicator 'k/G'`, readers/cap2det_reader.py:201-211) and never synchronises gradients
(TF parameter server, train_wsod.sh:46-88); the synchronous mean used here is what its
`SyncReplicasOptimizer` option computes (train/trainer.py:90-94).
"""
import os

import torch
import torch.distributed as dist


def world_info():
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def collectives_on():
  """True when the reducers must issue their collectives: more than one rank — or a process
  group of ONE rank with C2D_FORCE_ALLREDUCE=1, the rehearsal of the RCCL path (communicator
  setup, the all-reduce on RCCL's stream beside the step's compute / side / look-ahead streams,
  the stream joins around it) on a box with a single GPU.  A one-rank sum is the identity, so
  the step's results must not change."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return dist.get_world_size() > 1 or os.environ.get("C2D_FORCE_ALLREDUCE") == "1"


from cap2det_amd.train.gpu_count import count_visible_gpus  # noqa: E402,F401  (torch-free module)


def shard_range(num_items, rank, world):
  """Contiguous, disjoint, exhaustive shard [lo, hi) of `num_items` for `rank`."""
  if not (0 <= rank < world):
    raise ValueError("rank %d outside world %d" % (rank, world))
  base, rem = divmod(num_items, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def allreduce_bucket(flat, group=None):
  """Sums `flat` (a contiguous 1-D slice of the flat gradient buffer) over all ranks in place.
  Returns the factor the optimiser must scale the sum by (1 / world size)."""
  _, world = world_info()
  if collectives_on():
    if not flat.is_contiguous():
      raise ValueError("gradient bucket must be contiguous")
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
  return 1.0 / world


# bench.py --gpus N: event pairs around every finish() (GPU time the compute stream spends between
# "all gradients queued" and "all collectives done": what the overlap did NOT hide), when enabled
exposed_events = None


def _finish_begin():
  if exposed_events is None:
    return None
  ev = torch.cuda.Event(enable_timing=True)
  ev.record()
  return ev


def _finish_end(ev0):
  if ev0 is not None:
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    exposed_events.append((ev0, ev1))


class OverlappedReducer(object):
  """Two-bucket reduction of the flat gradient buffer, the big one under the tail of backward.

  The flat gradient bucket is laid out [first-stage trainable part | second stage | heads].
  Everything from the second stage on is final as soon as the second-stage backward has run,
  while the ROI-crop backward and the first-stage layers (Mixed_4e in the shipped configs)
  still have ~1 ms of kernels to go: `start_tail()` launches the all-reduce of that suffix
  (>= 85 % of the bytes) asynchronously — RCCL runs it on its own stream after the gradient
  kernels already enqueued — and `finish()` reduces the small prefix and waits for both.
  With one rank both are no-ops (unless C2D_FORCE_ALLREDUCE=1: `collectives_on`)."""

  def __init__(self, flat, split, group=None):
    self.flat, self.split, self.group = flat, int(split), group
    self._work = None
    _, self.world = world_info()
    self.on = collectives_on()

  def start_tail(self):
    if self.on and self.split < self.flat.numel():
      self._work = dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, group=self.group,
                                   async_op=True)

  def finish(self):
    if self.on:
      ev0 = _finish_begin()
      head = self.flat[:self.split] if self._work is not None else self.flat
      if head.numel():
        dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)
      if self._work is not None:
        self._work.wait()
        self._work = None
      _finish_end(ev0)
    return 1.0 / self.world


class BlockReducer(object):
  """Per-block reduction of the flat gradient bucket under the backward pass.

  The bucket is laid out in network order, [Mixed_4e | Mixed_5a | Mixed_5b | Mixed_5c | heads], and
  the backward pass finishes it from the END: heads, Mixed_5c, 5b, 5a, and Mixed_4e last.  `cuts`
  are the ascending offsets of those ranges inside `flat` (cuts[0] = 0, cuts[-1] = flat.numel());
  `start(i)` launches the asynchronous all-reduce of range i as soon as the caller has enqueued the
  last kernel that writes it — from inside whatever stream is current, so that the collective
  (RCCL runs it on its own stream) is ordered behind that stream's work and nobody else waits —
  and `finish()` reduces the ranges that were never started (the Mixed_4e prefix) and waits for
  all of them.  At 3.4 ms per bf16 step the one 24 MB suffix of OverlappedReducer no longer hides
  under the ≈0.6 ms of ROI-crop / Mixed_4e backward behind it; three ranges of 5-10 MB each start
  0.5-2 ms earlier (≈0.1 ms per range at the per-link xGMI rate of an 8-GPU ring).
  With one rank every call is a no-op (unless C2D_FORCE_ALLREDUCE=1: `collectives_on`)."""

  def __init__(self, flat, cuts, group=None):
    cuts = [int(c) for c in cuts]
    if cuts[0] != 0 or cuts[-1] != flat.numel() or any(a >= b for a, b in zip(cuts, cuts[1:])):
      raise ValueError("cuts must rise from 0 to the bucket size: %r" % (cuts,))
    self.flat, self.cuts, self.group = flat, cuts, group
    self._works = {}
    _, self.world = world_info()
    self.on = collectives_on()

  def num_ranges(self):
    return len(self.cuts) - 1

  def start(self, i):
    if self.on and i not in self._works:
      self._works[i] = dist.all_reduce(self.flat[self.cuts[i]:self.cuts[i + 1]],
                                       op=dist.ReduceOp.SUM, group=self.group, async_op=True)

  def finish(self):
    if self.on:
      ev0 = _finish_begin()
      # what is left, as few collectives as possible: runs of adjacent unstarted ranges
      i, n = 0, self.num_ranges()
      while i < n:
        if i in self._works:
          i += 1
          continue
        j = i
        while j + 1 < n and (j + 1) not in self._works:
          j += 1
        dist.all_reduce(self.flat[self.cuts[i]:self.cuts[j + 1]], op=dist.ReduceOp.SUM,
                        group=self.group)
        i = j + 1
      for w in self._works.values():
        w.wait()
      self._works = {}
      _finish_end(ev0)
    return 1.0 / self.world
