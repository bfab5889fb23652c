"""Micro-benchmark of the conv kernels on the second-stage shapes (N=2000 ROIs)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
SHAPES = [  # n, ih, cin, cout, k, stride
    (2000, 7, 576, 128, 1, 1), (2000, 7, 576, 192, 1, 1), (2000, 7, 192, 256, 3, 1),
    (2000, 7, 128, 192, 3, 2), (2000, 7, 256, 256, 3, 2), (2000, 4, 1024, 352, 1, 1),
    (2000, 4, 1024, 192, 1, 1), (2000, 4, 192, 320, 3, 1), (2000, 4, 224, 224, 3, 1),
    (2000, 4, 1024, 128, 1, 1), (1, 32, 128, 192, 3, 1), (1, 32, 576, 96, 1, 1),
    (2000, 1, 1024, 112, 1, 1)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
for (n, ih, cin, cout, k, s) in SHAPES:
    oh = -(-ih // s)
    x = torch.randn(n * ih * ih, cin, device=dev)
    w = torch.randn(k * k, cin, cout, device=dev) / (k * k * cin) ** 0.5
    wt = torch.empty(k * k, cout, cin, device=dev); ops.transpose_taps(w, wt, k * k, cin, cout)
    y = torch.empty(n * oh * oh, cout, device=dev); dy = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.zeros_like(w)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    fl = 2.0 * n * oh * oh * cin * cout * k * k
    res = []
    if which in ("all", "fwd"):
        t = timeit(lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, n, ih, ih, cin, cout, k, k, s, True)); res.append("fwd %7.1f us %6.1f TF" % (t * 1e3, fl / t / 1e9)); tot["fwd"] += t
    if which in ("all", "dgrad"):
        t = timeit(lambda: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, s, False)); res.append("dgrad %7.1f us %6.1f TF" % (t * 1e3, fl / t / 1e9)); tot["dgrad"] += t
    if which in ("all", "wgrad"):
        t = timeit(lambda: ops.conv_wgrad(x, cin, 0, dy, cout, 0, dw, n, ih, ih, cin, cout, k, k, s)); res.append("wgrad %7.1f us %6.1f TF" % (t * 1e3, fl / t / 1e9)); tot["wgrad"] += t
    print("n=%4d %2dx%-2d cin=%4d cout=%3d k=%d s=%d | %s" % (n, ih, ih, cin, cout, k, s, " | ".join(res)))
print("sum ms:", {k: round(v, 3) for k, v in tot.items()})
