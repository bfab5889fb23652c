"""ctypes loader for libcap2det_hip.so (the C-ABI of include/cap2det_hip.h).

There is NO CPU fallback: if the shared library is missing or a symbol is absent the
product path raises.  `import torch` happens first so that the HIP runtime the library binds
to (soname libamdhip64.so.7) is the one PyTorch already loaded — device pointers and streams
are then shared between torch and the kernels.

Argument types are taken from the declarations in include/cap2det_hip.h itself, so the
binding cannot drift from the header.
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# C2D_LIB: tools-only override (the diagnostic trace build); the product always loads the in-tree
# library.
LIB_PATH = os.environ.get("C2D_LIB") or os.path.join(_HERE, "csrc", "libcap2det_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cap2det_hip.h")

_lib = None
_SIGS = None


class Cap2DetHipError(RuntimeError):
  """Raised when a C-ABI call returns a negative C2D_ERR_* code."""


def _ctype(decl):
  decl = decl.strip()
  if "*" in decl:
    return ctypes.c_void_p
  base = re.sub(r"\b[A-Za-z_]\w*$", "", decl).strip() if " " in decl else decl
  base = base.replace("const", "").strip()
  return {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong,
          "unsigned long long": ctypes.c_ulonglong, "int32_t": ctypes.c_int32,
          "unsigned int": ctypes.c_uint, "void": None}[base]


def header_signatures():
  """Parses `int c2d_xxx(...)` declarations of the header -> {name: (restype, [argtypes])}."""
  global _SIGS
  if _SIGS is not None:
    return _SIGS
  with open(HEADER_PATH) as f:
    text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
  sigs = {}
  for m in re.finditer(r"(const char\*|unsigned long long|unsigned int|long long|int)\s+(c2d_\w+)"
                       r"\s*\(([^)]*)\)\s*;", text):
    ret, name, params = m.group(1), m.group(2), m.group(3).strip()
    args = [] if params in ("void", "") else [_ctype(p) for p in params.split(",")]
    restype = {"int": ctypes.c_int, "long long": ctypes.c_longlong,
               "unsigned int": ctypes.c_uint,
               "unsigned long long": ctypes.c_ulonglong}.get(ret, ctypes.c_char_p)
    sigs[name] = (restype, args)
  _SIGS = sigs
  return sigs


def header_abi_version():
  """`#define C2D_ABI_VERSION n` of the header."""
  with open(HEADER_PATH) as f:
    return int(re.search(r"#define\s+C2D_ABI_VERSION\s+(\d+)", f.read()).group(1))


def load():
  """Loads (once) and returns the ctypes handle; raises if the extension is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cap2det_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; "
        "g.build()'` (or `make -C cap2det_amd/csrc`). There is no CPU fallback." % LIB_PATH)
  import torch  # noqa: F401  (loads the HIP runtime first, see module docstring)
  lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
  for name, (restype, argtypes) in header_signatures().items():
    fn = getattr(lib, name)  # AttributeError => header/library mismatch: fail loudly
    fn.restype = restype
    fn.argtypes = argtypes
  want = header_abi_version()
  got = lib.c2d_version()
  if got != want:
    raise ImportError("cap2det_amd: %s was built for ABI %d, include/cap2det_hip.h declares %d — "
                      "rebuild it (`make -C cap2det_amd/csrc`)" % (LIB_PATH, got, want))
  _lib = lib
  return lib


def check(code, what):
  if code != 0:
    msg = load().c2d_error_string(int(code)).decode()
    raise Cap2DetHipError("%s failed: %s (%d)" % (what, msg, code))


# The step-plan recorder (cap2det_amd/step_plan.py) while a step is being recorded, else None: every
# call below is then also appended to the plan (after it ran: a recorded step is a real step).
recorder = None
recorder_thread = None      # only calls of the thread that records belong to the plan (an input
                            # thread queues its reader kernels through this module at the same time)


def call(name, *args):
  """Calls C-ABI function `name`; raises Cap2DetHipError on a non-zero return."""
  check(getattr(load(), name)(*args), name)
  if recorder is not None and threading.get_ident() == recorder_thread:
    recorder.add_call(name, args)


def call_sym(name, *args):
  """`call` for the few entry points that take per-step scalars (dropout key, learning rate): an
  argument may be a step_plan.Sym — its value now, its binding slot in a recorded plan."""
  check(getattr(load(), name)(*[getattr(a, "value", a) if a.__class__.__name__ == "Sym" else a
                                for a in args]), name)
  if recorder is not None and threading.get_ident() == recorder_thread:
    recorder.add_call(name, args)
