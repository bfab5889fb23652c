"""Step plans without a GPU (csrc/plan.hip, cap2det_amd/step_plan.py): the node list, the typed thunks
generated from the header, argument binding and the error path — with entry points whose argument
checks fail before anything touches a device.  Replay against real steps: tests/test_gpu_model.py::
test_plan_step_equals_eager_step."""
import ctypes

import pytest

from cap2det_amd import _lib


def _plan():
  lib = _lib.load()
  return lib, ctypes.c_void_p(lib.c2d_plan_create())


def _add(lib, plan, name, words, kinds=None, slots=None):
  n = len(words)
  vals = (ctypes.c_longlong * max(n, 1))(*words)
  k = (ctypes.c_uint8 * max(n, 1))(*(kinds or [0] * n))
  s = (ctypes.c_int * max(n, 1))(*(slots or [0] * n))
  return lib.c2d_plan_add_call(plan, name, n, vals, k, s)


def test_every_int_entry_point_has_a_thunk():
  lib, plan = _plan()
  try:
    for name, (restype, argtypes) in _lib.header_signatures().items():
      if restype is not ctypes.c_int or name.startswith("c2d_plan_"):
        continue
      assert _add(lib, plan, name.encode(), [0] * len(argtypes)) == 0, name
      assert _add(lib, plan, name.encode(), [0] * (len(argtypes) + 1)) == -1, name   # wrong arity
    assert _add(lib, plan, b"c2d_no_such_entry_point", []) == -2
    assert lib.c2d_plan_size(plan) > 100
  finally:
    lib.c2d_plan_destroy(plan)


def test_replay_stops_at_the_first_failing_node_and_applies_bindings():
  lib, plan = _plan()
  try:
    # c2d_split3_bf16(src, planes, plane_stride, n, stream): n comes from binding slot 1, src from
    # slot 0 (+16 bytes); with n = 0 the argument check fails, with an odd pointer too
    assert _add(lib, plan, b"c2d_f32x9_unbind", [0]) == 0                       # succeeds anywhere
    assert _add(lib, plan, b"c2d_split3_bf16", [16, 4096, 8, 0, 0], [1, 0, 0, 2, 0], [0, 0, 0, 1, 0]) == 0
    failed = ctypes.c_int(-7)
    assert lib.c2d_plan_replay(plan, None, 0, ctypes.byref(failed)) == -1       # plan not finished
    assert lib.c2d_plan_finish(plan, None) == 0
    assert _add(lib, plan, b"c2d_f32x9_unbind", [0]) == -1                      # closed
    binds = (ctypes.c_longlong * 2)(1 << 20, 0)
    assert lib.c2d_plan_replay(plan, binds, 1, ctypes.byref(failed)) == -1      # too few bindings
    assert lib.c2d_plan_replay(plan, binds, 2, ctypes.byref(failed)) == -1 and failed.value == 1
  finally:
    lib.c2d_plan_destroy(plan)


def test_recorder_turns_arguments_into_words():
  from cap2det_amd import step_plan
  plan = step_plan.StepPlan.__new__(step_plan.StepPlan)      # (no torch stream hooks on a CPU box)
  plan.lib = _lib.load()
  plan.handle = ctypes.c_void_p(plan.lib.c2d_plan_create())
  plan.sigs = _lib.header_signatures()
  plan.keep, plan.ranges, plan.slot_names, plan.slot_kind = [], [(4096, 8192, 0)], ["ex.image"], {"ex.image": "tensor"}
  plan.tensor_meta, plan.nodes, plan.calls, plan.finished = {}, [], 0, False
  table = (ctypes.c_longlong * 2)(7, 9)
  plan.add_call("c2d_adagrad_step_multi", (4100, None, 123, 2, table, table, table, table,
                                            step_plan.Sym("lr", 0.5), 1.0, None, None))
  (kind, name, args), = plan.nodes
  assert (kind, name) == ("call", "c2d_adagrad_step_multi")
  assert args[0] == (1, "ex.image", 4)             # pointer into the bound tensor: offset 4
  assert args[1] == (0, None, 0) and args[2] == (0, None, 123)
  assert args[8] == (2, "lr", None)                # the symbolic scalar
  assert args[9][2] == 0x3f800000                  # 1.0f as a word
  assert plan.structure()[0][2][4] == (0, None, ("host", bytes(table)))
  plan.lib.c2d_plan_destroy(plan.handle)
  plan.handle = None
