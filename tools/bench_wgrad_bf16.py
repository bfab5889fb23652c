"""Filter-gradient kernels on the second-stage conv shapes: fp32 operands (fp32 MFMA) next to bf16
operands (bf16 MFMA; with C2D_TUNE=1 C2D_WGRAD_BF16_MFMA=0: the widening fp32-MFMA kernels)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
SHAPES = [(2000, 7, 576, 128, 1, 1), (2000, 7, 576, 192, 1, 1), (2000, 7, 192, 256, 3, 1),
          (2000, 7, 256, 256, 3, 2), (2000, 7, 128, 192, 3, 2), (2000, 4, 1024, 352, 1, 1),
          (2000, 4, 1024, 192, 1, 1), (2000, 4, 192, 320, 3, 1), (2000, 4, 160, 224, 3, 1),
          (2000, 4, 224, 224, 3, 1), (2000, 4, 1024, 128, 1, 1)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
tot = {}
for (n, ih, cin, cout, k, st) in SHAPES:
    oh = -(-ih // st)
    fl = 2.0 * n * oh * oh * cin * cout * k * k
    res = []
    for dt in ((torch.bfloat16,) if os.environ.get("C2D_BENCH_BF16_ONLY") else (torch.float32, torch.bfloat16)):
        x = torch.randn(n * ih * ih, cin, device=dev).to(dt)
        dy = torch.randn(n * oh * oh, cout, device=dev).to(dt)
        dw = torch.zeros(k * k, cin, cout, device=dev)
        t = timeit(lambda: ops.conv_wgrad(x, cin, 0, dy, cout, 0, dw, n, ih, ih, cin, cout, k, k, st))
        name = "f32" if dt == torch.float32 else "bf16"
        res.append("%s %6.1f us %6.1f TF" % (name, t * 1e3, fl / t / 1e9))
        tot[name] = tot.get(name, 0) + t
    print("n=%4d %dx%d cin=%4d cout=%3d k=%d s=%d | %s" % (n, ih, ih, cin, cout, k, st, " | ".join(res)))
print("sum ms:", {k: round(v, 3) for k, v in tot.items()})
