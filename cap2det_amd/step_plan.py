"""Step plans: record the C-ABI call list of one training step while it runs eagerly, replay it with
ONE native call (csrc/plan.hip; the reference's counterpart is the single session.run of
train/trainer.py:141-146).

Recording hooks two things and nothing else:
  * `_lib.call` — every entry-point call of the step, with its argument words;
  * the stream plumbing the step uses from PyTorch — `torch.cuda.Event()`, `Event.record()`,
    `Stream.wait_event()`, `Stream.wait_stream()` — so that the plan carries the same forks and joins
    on the same (persistent) HIP streams.
What may differ between replays is declared up front: `bind_tensor` (an input tensor: pointer
arguments inside it are re-based at replay) and `Sym` scalars (dropout key, learning rate).
A wait for an event that was recorded BEFORE the recording started (the eager schedule's cross-step
events) is not recorded: c2d_plan_finish orders every other stream of the plan behind the main
stream at the start of a replay and joins it at the end.
"""
import ctypes
import threading

import torch

from cap2det_amd import _lib


class Sym(object):
  """A scalar argument that changes from replay to replay: `value` now, binding slot `name`."""
  __slots__ = ("name", "value")

  def __init__(self, name, value):
    self.name, self.value = name, value

  def __float__(self):
    return float(self.value)

  def __int__(self):
    return int(self.value)

  __index__ = __int__


def sym_like(orig, converted):
  """`converted` (what a wrapper makes of the argument `orig`), still symbolic if `orig` was."""
  return Sym(orig.name, converted) if isinstance(orig, Sym) else converted


class _RecEvent(object):
  """torch.cuda.Event created while a step is recorded: the real event plus its index in the plan."""

  def __init__(self, rec, *args, **kwargs):
    self.real = _RealEvent(*args, **kwargs)
    self.rec, self.index = rec, rec.new_event()

  def record(self, stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    self.real.record(s)
    if _recorder() is self.rec:
      self.rec.add_record(self.index, s.cuda_stream)

  def __getattr__(self, k):        # synchronize / query / elapsed_time / wait
    return getattr(self.real, k)


_RealEvent = torch.cuda.Event
_real_wait_event = torch.cuda.Stream.wait_event
_real_wait_stream = torch.cuda.Stream.wait_stream


def _recorder():
  """The active recorder if the CALLING thread is the one recording, else None."""
  rec = _lib.recorder
  if rec is not None and threading.get_ident() == _lib.recorder_thread:
    return rec
  return None


def _event_factory(*args, **kwargs):
  rec = _recorder()
  if rec is None:
    return _RealEvent(*args, **kwargs)
  return _RecEvent(rec, *args, **kwargs)


def _wait_event(self, event):
  if isinstance(event, _RecEvent):
    _real_wait_event(self, event.real)
    rec = _recorder()
    if rec is not None and rec is event.rec:
      rec.add_wait(self.cuda_stream, event.index)
    return
  _real_wait_event(self, event)
  rec = _recorder()
  if rec is not None:
    rec.external_waits += 1


def _wait_stream(self, stream):
  rec = _recorder()
  if rec is None:
    return _real_wait_stream(self, stream)
  ev = _RecEvent(rec)
  ev.record(stream)
  _wait_event(self, ev)


_installed = False


def install():
  """Routes torch.cuda.Event / Stream.wait_event / Stream.wait_stream through the recorder hooks
  (plain pass-through while nothing is recording).  Idempotent."""
  global _installed
  if _installed:
    return
  torch.cuda.Event = _event_factory
  torch.cuda.Stream.wait_event = _wait_event
  torch.cuda.Stream.wait_stream = _wait_stream
  _installed = True


_KIND = {ctypes.c_float: "f"}


class StepPlan(object):
  """Recorder while `recording()`, then the finished plan: `replay(tensors, scalars)`."""

  def __init__(self):
    install()
    self.lib = _lib.load()
    self.handle = ctypes.c_void_p(self.lib.c2d_plan_create())
    if not self.handle.value:
      raise MemoryError("c2d_plan_create")
    self.sigs = _lib.header_signatures()
    self.keep = []                 # host arrays / structs whose addresses the plan holds
    self.ranges = []               # (lo, hi, slot) of the bound tensors
    self.slot_names = []           # slot -> name
    self.slot_kind = {}            # name -> "tensor" | "float" | "int"
    self.tensor_meta = {}          # name -> (shape, dtype)
    self.external_waits = 0
    self.num_events = 0
    self.calls = 0
    self.nodes = []                # mirror of the recorded nodes (structure())
    self.finished = False
    self._bind_buf = None

  def __del__(self):
    try:
      if self.handle and self.handle.value:
        self.lib.c2d_plan_destroy(self.handle)
    except Exception:   # noqa: BLE001 -- interpreter shutdown
      pass

  # -- declaring what varies ------------------------------------------------------
  def _slot(self, name, kind):
    if name in self.slot_kind:
      assert self.slot_kind[name] == kind, name
      return self.slot_names.index(name)
    self.slot_kind[name] = kind
    self.slot_names.append(name)
    return len(self.slot_names) - 1

  def bind_tensor(self, name, t):
    """Pointer arguments that point into `t` are re-based on the tensor given for `name` at replay."""
    assert t.is_cuda and t.is_contiguous()
    slot = self._slot(name, "tensor")
    lo = t.data_ptr()
    self.ranges.append((lo, lo + max(t.numel() * t.element_size(), 1), slot))
    self.tensor_meta[name] = (tuple(t.shape), t.dtype)

  # -- recording --------------------------------------------------------------------
  def recording(self):
    plan = self

    class _Scope(object):
      def __enter__(self):
        assert _lib.recorder is None and not plan.finished
        plan.main_stream = torch.cuda.current_stream().cuda_stream
        _lib.recorder_thread = threading.get_ident()
        _lib.recorder = plan
        return plan

      def __exit__(self, exc_type, exc, tb):
        _lib.recorder = None
        if exc_type is None:
          _lib.check(plan.lib.c2d_plan_finish(plan.handle, plan.main_stream), "c2d_plan_finish")
          plan.finished = True
          plan._bind_buf = (ctypes.c_longlong * max(len(plan.slot_names), 1))()
        return False

    return _Scope()

  def new_event(self):
    self.num_events += 1
    return self.num_events - 1

  def add_record(self, event, stream):
    self.nodes.append(("record", event, stream))
    _lib.check(self.lib.c2d_plan_add_event_record(self.handle, event, stream), "c2d_plan_add_event_record")

  def add_wait(self, stream, event):
    self.nodes.append(("wait", stream, event))
    _lib.check(self.lib.c2d_plan_add_stream_wait(self.handle, stream, event), "c2d_plan_add_stream_wait")

  def add_call(self, name, args):
    argtypes = self.sigs[name][1]
    n = len(args)
    assert n == len(argtypes), name
    vals = (ctypes.c_longlong * max(n, 1))()
    kinds = (ctypes.c_uint8 * max(n, 1))()
    slots = (ctypes.c_int * max(n, 1))()
    for i, (a, t) in enumerate(zip(args, argtypes)):
      if isinstance(a, Sym):
        kind = "float" if t is ctypes.c_float else "int"
        slots[i] = self._slot(a.name, kind)
        kinds[i] = 2
        continue
      if t is ctypes.c_float:
        vals[i] = ctypes.c_uint32.from_buffer_copy(ctypes.c_float(a)).value
      elif t is ctypes.c_void_p:
        if a is None:
          v = 0
        elif isinstance(a, int):
          v = a
        elif isinstance(a, (bytes, bytearray)):
          buf = ctypes.create_string_buffer(bytes(a))
          self.keep.append(buf)
          v = ctypes.addressof(buf)
        else:                       # ctypes array / structure passed by pointer: keep it alive
          self.keep.append(a)
          if isinstance(a, ctypes.c_void_p):
            v = a.value or 0
          elif isinstance(a, ctypes._Pointer):
            v = ctypes.cast(a, ctypes.c_void_p).value or 0
          elif hasattr(a, "_obj"):                       # ctypes.byref(x)
            v = ctypes.addressof(a._obj)
          else:
            try:
              v = ctypes.addressof(a)
            except TypeError:
              raise TypeError("step plan: argument %d of %s is a %r" % (i, name, type(a)))
        for lo, hi, slot in self.ranges:
          if lo <= v < hi:
            kinds[i], slots[i], v = 1, slot, v - lo
            break
        vals[i] = v
      else:
        v = int(a)
        vals[i] = v - (1 << 64) if v >= (1 << 63) else v
    _lib.check(self.lib.c2d_plan_add_call(self.handle, name.encode(), n, vals, kinds, slots),
               "c2d_plan_add_call(%s)" % name)
    self.calls += 1
    self.nodes.append(("call", name, tuple(
        (int(kinds[i]), self.slot_names[slots[i]] if kinds[i] else None,
         int(vals[i]) if kinds[i] != 2 else None) for i in range(n))))

  # -- replay ---------------------------------------------------------------------------
  def replay(self, tensors, scalars):
    """tensors: {name: device tensor} for every bound tensor (same shape / dtype as recorded);
    scalars: {name: value} for every Sym."""
    assert self.finished
    buf = self._bind_buf
    for i, name in enumerate(self.slot_names):
      kind = self.slot_kind[name]
      if kind == "tensor":
        t = tensors[name]
        if (tuple(t.shape), t.dtype) != self.tensor_meta[name] or not t.is_contiguous():
          raise ValueError("step plan: tensor %r changed shape / dtype / layout" % name)
        buf[i] = t.data_ptr()
      elif kind == "float":
        buf[i] = ctypes.c_uint32.from_buffer_copy(ctypes.c_float(scalars[name])).value
      else:
        v = int(scalars[name])
        buf[i] = v - (1 << 64) if v >= (1 << 63) else v
    failed = ctypes.c_int(-1)
    rc = self.lib.c2d_plan_replay(self.handle, buf, len(self.slot_names), ctypes.byref(failed))
    if rc != 0:
      raise _lib.Cap2DetHipError("c2d_plan_replay failed at node %d: %s (%d)"
                                 % (failed.value, self.lib.c2d_error_string(int(rc)).decode(), rc))

  def structure(self):
    """The recorded nodes with bindings by NAME: equal for two recordings of the same step shape
    (constants that are addresses of host descriptor arrays compare by the arrays' contents)."""
    host = {}
    for obj in self.keep:
      obj = getattr(obj, "_obj", obj)          # (ctypes.byref(x) -> x)
      try:
        host[ctypes.addressof(obj)] = bytes(obj)
      except TypeError:
        pass
    out = []
    for node in self.nodes:
      if node[0] != "call":
        out.append(node)
        continue
      args = tuple((k, sl, ("host", host[v]) if (k == 0 and v in host) else v) for k, sl, v in node[2])
      out.append(("call", node[1], args))
    return out

  def size(self):
    return int(self.lib.c2d_plan_size(self.handle))
