"""bf16 igemm (fwd / dgrad) and filter gradient on every distinct second-stage conv shape at
N = 2000 ROIs.  Prints per call: time, TFLOP/s, the kernel instance dispatched.  Tuning hooks
(C2D_TUNE=1 C2D_IGEMM_CFG=2|3|6|7 block tile, C2D_RING_BK / C2D_RING_D stage and ring depth) select the
variant; tools/sweep_ring.sh runs them side by side."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
SHAPES = [(7, 576, 128, 1, 1), (7, 128, 192, 3, 2), (7, 576, 192, 1, 1), (7, 192, 256, 3, 1),
          (7, 256, 256, 3, 2), (4, 1024, 352, 1, 1), (4, 1024, 192, 1, 1), (4, 192, 320, 3, 1),
          (4, 1024, 160, 1, 1), (4, 160, 224, 3, 1), (4, 224, 224, 3, 1), (4, 192, 224, 3, 1),
          (4, 1024, 128, 1, 1)]
# how often each shape occurs in one step of Mixed_5a-c
COUNT = {(4, 1024, 352, 1, 1): 2, (4, 1024, 192, 1, 1): 3, (4, 192, 320, 3, 1): 2,
         (4, 224, 224, 3, 1): 2, (4, 1024, 128, 1, 1): 2}
n = 2000
if os.environ.get("C2D_BENCH_SHAPES"):        # e.g. "5,7": only these rows of SHAPES (profiling)
    SHAPES = [SHAPES[int(i)] for i in os.environ["C2D_BENCH_SHAPES"].split(",")]
ONLY = os.environ.get("C2D_BENCH_CALLS", "").split(",") if os.environ.get("C2D_BENCH_CALLS") else None
ITERS = int(os.environ.get("C2D_BENCH_ITERS", "20"))
what = sys.argv[1] if len(sys.argv) > 1 else "igemm"
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
PEAK = 157.3 if DT == torch.float32 else 2500.0
def timeit(fn, iters=ITERS):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
tot_t, tot_f = {}, {}
dt = DT
for (ih, cin, cout, k, st) in SHAPES:
    oh = -(-ih // st)
    fl = 2.0 * n * oh * oh * cin * cout * k * k
    x = torch.randn(n * ih * ih, cin, device=dev).to(dt)
    w = (torch.randn(k * k, cin, cout, device=dev) / (k * k * cin) ** 0.5).to(dt)
    wt = w.permute(0, 2, 1).contiguous()
    y = torch.empty(n * oh * oh, cout, device=dev, dtype=dt); dy = torch.randn(n * oh * oh, cout, device=dev).to(dt)
    dx = torch.empty_like(x); dw = torch.zeros(k * k, cin, cout, device=dev)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    calls = {"fwd": lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, n, ih, ih, cin, cout, k, k, st, True),
             "dgrad": lambda: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, st, False),
             "wgrad": lambda: ops.conv_wgrad(x, cin, 0, dy, cout, 0, dw, n, ih, ih, cin, cout, k, k, st)}
    line = "%dx%d cin=%4d cout=%3d k=%d s=%d |" % (ih, ih, cin, cout, k, st)
    for name in (("fwd", "dgrad") if what == "igemm" else ("wgrad",)):
        if ONLY and name not in ONLY:
            continue
        t = timeit(calls[name])
        inst = ops.last_dispatch()
        c = COUNT.get((ih, cin, cout, k, st), 1)
        tot_t[name] = tot_t.get(name, 0) + c * t; tot_f[name] = tot_f.get(name, 0) + c * fl
        line += " %s %6.1f us %6.1f TF %s |" % (name, t * 1e3, fl / t / 1e9, inst[0].replace("_kernel", "") if inst else "?")
    print(line)
for k_ in tot_t:
    print("per step %s: %.3f ms, %.1f TF (%.3f of %.0f)" % (k_, tot_t[k_], tot_f[k_] / tot_t[k_] / 1e9, tot_f[k_] / tot_t[k_] / 1e9 / PEAK, PEAK))
