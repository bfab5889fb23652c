"""`dataset.prefetch(options.prefetch_buffer_size)` of the reference's input pipeline
(readers/cap2det_reader.py:266) for a GPU consumer.

A daemon thread pulls batches from the input function under a COPY STREAM of its own, so the
reader's uploads (pinned host memory -> HBM) and its flip / resize / pad kernels are queued beside
whatever training step the compute stream is running, never in front of it; every batch carries the
event that marks its last upload / kernel (`batch["_ready"]`).  The consumer makes the stream that
READS a batch wait for that event (Trainer.train: the compute stream for the current batch, the
look-ahead stream for the next one) — the copy stream itself never waits for the compute stream:
the caching allocator keeps the blocks of the two streams apart, and `adopt()` records the consumer
streams on the batch tensors so that a block is not handed out again while a step still reads it.
"""
import queue
import threading

import torch

_END = object()


class DevicePrefetcher(object):
  """Iterator over `batches` (any iterable of example dicts) that stays `depth` batches ahead."""

  def __init__(self, batches, device, depth=2):
    self.device = torch.device(device)
    self.cuda = self.device.type == "cuda"
    self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
    self._q = queue.Queue(maxsize=max(int(depth), 1))
    self._stop = threading.Event()
    self._it = iter(batches)
    # (an iterator handed in by the caller stays the caller's: close() ends only what iter() made here)
    self._own_iterator = self._it is not batches
    self._thread = threading.Thread(target=self._run, name="c2d-input-prefetch", daemon=True)
    self._thread.start()

  def _put(self, item):
    while not self._stop.is_set():
      try:
        self._q.put(item, timeout=0.1)
        return True
      except queue.Full:
        continue
    return False

  def _run(self):
    try:
      if self.cuda:
        torch.cuda.set_device(self.device)
      while not self._stop.is_set():
        if self.cuda:
          with torch.cuda.stream(self.stream):
            batch = next(self._it, _END)
            ready = None
            if batch is not _END:
              ready = torch.cuda.Event()
              ready.record(self.stream)
        else:
          batch, ready = next(self._it, _END), None
        if batch is _END:
          break
        batch = dict(batch)
        batch["_ready"] = ready
        if not self._put(batch):
          return
      self._put(_END)
    except BaseException as e:   # noqa: BLE001 -- handed to the consumer, which re-raises it
      self._put(e)

  def __iter__(self):
    return self

  def __next__(self):
    item = self._q.get()
    if item is _END:
      self._q.put(_END)          # (a second next() ends as well)
      raise StopIteration
    if isinstance(item, BaseException):
      self._q.put(_END)
      raise item
    return item

  def close(self):
    self._stop.set()
    try:
      while True:
        self._q.get_nowait()
    except queue.Empty:
      pass
    self._thread.join(timeout=5.0)
    # (an iterator made HERE from the caller's iterable may own resources: end it, once the thread
    #  that was running it has stopped; an iterator the caller handed in is the caller's to close or
    #  to keep pulling from — the batches this prefetcher had already taken from it are dropped)
    closer = getattr(self._it, "close", None) if self._own_iterator else None
    if closer is not None and not self._thread.is_alive():
      try:
        closer()
      except Exception:   # noqa: BLE001
        pass


def adopt(batch, *streams):
  """Marks the device tensors of a prefetched batch as used by `streams` (the caching allocator then
  defers re-use of their blocks until those streams have passed this point)."""
  for v in batch.values():
    if isinstance(v, torch.Tensor) and v.is_cuda:
      for s in streams:
        if s is not None:
          v.record_stream(s)
