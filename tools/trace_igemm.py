"""Per-block timeline of one igemm launch from the diagnostic build (make -C cap2det_amd/csrc trace):
start/end (s_memrealtime, 100 MHz), shader cycles (s_memtime), XCC / HW id of every block.

  C2D_LIB=cap2det_amd/csrc/libcap2det_hip_trace.so python tools/trace_igemm.py fwd 2000 4 192 320 3 1
"""
import ctypes, os, sys
import numpy as np, torch
os.environ.setdefault("C2D_TUNE", "1")
sys.path.insert(0, ".")
from cap2det_amd import _lib, hip_ops as ops
dev = "cuda:0"
which = sys.argv[1]; n, ih, cin, cout, k, s = [int(v) for v in sys.argv[2:8]]
oh = -(-ih // s)
x = torch.randn(n * ih * ih, cin, device=dev); w = torch.randn(k * k, cin, cout, device=dev) * 0.05
wt = torch.empty(k * k, cout, cin, device=dev); ops.transpose_taps(w, wt, k * k, cin, cout)
y = torch.empty(n * oh * oh, cout, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x)
trace = torch.zeros(8 * 8192 * 5, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
if os.environ.get("C2D_IGEMM_SK", "1") != "0":
    ws = torch.zeros(ops.conv_workspace_bytes(), dtype=torch.uint8, device=dev)
    ops.set_conv_workspace(ws)
def run():
    if which == "fwd": ops.conv_fwd(x, cin, 0, wt, None, None, y, cout, 0, n, ih, ih, cin, cout, k, k, s, True)
    else: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, s, False)
for _ in range(3): run()
torch.cuda.synchronize()
lib.c2d_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
run(); torch.cuda.synchronize()
lib.c2d_debug_set_trace(ctypes.c_void_p(0))
tall = trace.cpu().numpy().reshape(-1, 8)
w = tall[8192:]; w = w[w[:, 5] != 0]
if len(w):
    per = w[:, :5] / w[:, 5:6]
    names = ["wait loads + ds_write", "barrier 1", "cursor + issue loads", "ds_read + MFMA issue", "barrier 2"]
    print("per-wave shader cycles per slab (median over waves): " + "; ".join("%s %.0f" % (n, np.median(per[:, i])) for i, n in enumerate(names)) + "; sum %.0f" % np.median(per.sum(1)))
t = tall[:8192]
t = t[t[:, 1] != 0]
t0, t1, c0, c1, hw, tile, total, mask = [t[:, i] for i in range(8)]
total = np.maximum(total, 1)
base = t0.min()
start = (t0 - base) / 100.0; end = (t1 - base) / 100.0      # us
dur = end - start
clk = (c1 - c0) / np.maximum(t1 - t0, 1) * 100.0 / 1000.0   # GHz... s_memtime ticks per us / 1000
xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xF)
print("blocks %d  kernel span %.1f us  block dur us: min %.1f med %.1f max %.1f  slabs/block med %d"
      % (len(t), end.max(), dur.min(), np.median(dur), dur.max(), np.median(total)))
print("in-kernel clock GHz: med %.3f (min %.3f max %.3f)" % (np.median(clk), clk.min(), clk.max()))
print("distinct CUs %d; blocks per CU: %s" % (len(set(cu)), np.bincount(np.unique(cu, return_counts=True)[1])))
late = start > 0.05 * end.max()
print("blocks starting after 5%% of the span: %d (first such start %.1f us)" % (late.sum(), start[late].min() if late.any() else -1))
for lo in range(0, int(end.max()) + 1, max(int(end.max() / 12), 1)):
    hi = lo + max(int(end.max() / 12), 1)
    act = ((start < hi) & (end > lo)).sum()
    print("  t=%4d..%4d us: %4d blocks active" % (lo, hi, act))
us_per_slab = dur / np.maximum(total, 1)
print("us per slab: min %.2f med %.2f max %.2f" % (us_per_slab.min(), np.median(us_per_slab), us_per_slab.max()))
