"""In-kernel clock under the fp32-MFMA kernel and under the f32x9 kernel on the same GEMM
(MI355X_MICROARCH.md "DVFS give-back" item 6: delta s_memtime / delta s_memrealtime x 100 MHz, stamped
around a block after >= 2 s of back-to-back launches on random data).  Needs the diagnostic library:

  make -C cap2det_amd/csrc trace
  C2D_LIB=cap2det_amd/csrc/libcap2det_hip_trace.so python tools/x9_clock.py

GEMM: 2000 x 16 rows, K = 1408, N = 256 (the size of a 3x3 224->224 convolution over 2000 4x4 maps),
row-major, so every block does the same work."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import _lib, hip_ops as ops  # noqa: E402

dev = "cuda:0"
n, hw, cin, cout = 2000, 4, 1408, 256
x = torch.randn(n * hw * hw, cin, device=dev)
w = torch.randn(1, cout, cin, device=dev) / cin ** 0.5
y = torch.empty(n * hw * hw, cout, device=dev)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
run = lambda: ops.conv_fwd(x, cin, 0, w, sc, sh, y, cout, 0, n, hw, hw, cin, cout, 1, 1, 1, True)
lib = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(lib, "c2d_debug_set_ring_trace"):
  sys.exit("needs the diagnostic build (C2D_LIB=.../libcap2det_hip_trace.so)")
fl = 2.0 * n * hw * hw * cin * cout


def soak():
  t0 = time.time()
  while time.time() - t0 < 2.0:
    for _ in range(50):
      run()
    torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
    run()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 50


for mode in ("fp32", "x9"):
  if mode == "x9":
    keep = ops.x9_planes(w)
  ms = soak()
  if mode == "fp32":       # igemm_nt_kernel stamps: t[0..1] s_memrealtime, t[2..3] s_memtime (start, end)
    buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
    lib.c2d_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.c2d_debug_set_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 8)[:8192]
    t = t[t[:, 0] != 0].astype(np.float64)
    clk = (t[:, 3] - t[:, 2]) / (t[:, 1] - t[:, 0]) * 0.1
  else:                    # ring stamps: t[1] + t[2] + t[3] cycles, t[12] .. t[7] s_memrealtime
    buf = torch.zeros(16 * 65536, dtype=torch.int64, device=dev)
    lib.c2d_debug_set_ring_trace(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.c2d_debug_set_ring_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] != 0].astype(np.float64)
    clk = (t[:, 1] + t[:, 2] + t[:, 3]) / (t[:, 7] - t[:, 12]) * 0.1
  print("%-5s %7.1f us %6.1f TF  in-kernel clock GHz: median %.3f  p10 %.3f  p90 %.3f  (%d blocks)  %s"
        % (mode, ms * 1e3, fl / ms / 1e9, np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90),
           len(clk), ";".join(ops.last_dispatch())))
