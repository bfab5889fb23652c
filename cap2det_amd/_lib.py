"""ctypes loader for libcap2det_hip.so (the C-ABI of include/cap2det_hip.h).

There is NO CPU fallback: if the shared library is missing or a symbol is absent the
product path raises.  `import torch` happens first so that the HIP runtime the library binds
to (soname libamdhip64.so.7) is the one PyTorch already loaded — device pointers and streams
are then shared between torch and the kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcap2det_hip.so")

_lib = None


class Cap2DetHipError(RuntimeError):
  """Raised when a C-ABI call returns a negative C2D_ERR_* code."""


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
  lib.c2d_version.restype = ctypes.c_int
  lib.c2d_error_string.restype = ctypes.c_char_p
  lib.c2d_error_string.argtypes = [ctypes.c_int]
  _lib = lib
  return lib


def check(code, what):
  if code != 0:
    msg = load().c2d_error_string(int(code)).decode()
    raise Cap2DetHipError("%s failed: %s (%d)" % (what, msg, code))


def call(name, *args):
  """Calls C-ABI function `name` with already-converted ctypes-compatible args."""
  fn = getattr(load(), name)
  fn.restype = ctypes.c_int
  check(fn(*args), name)
