"""Does a power-of-two row stride (concat buffers with 1024 channels = 4096 B) hurt the conv kernels?"""
import sys, torch
sys.path.insert(0, ".")
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
n, ih = 2000, 4
big = torch.empty(160 * 1024 * 1024, device=dev)  # 640 MB scrubber
for (cin, cout, k) in [(1024, 352, 1), (192, 320, 3), (1024, 192, 1), (224, 224, 3)]:
    w = torch.randn(k * k, cin, cout, device=dev) * 0.05
    wt = torch.empty(k * k, cout, cin, device=dev); ops.transpose_taps(w, wt, k * k, cin, cout)
    for ldx_pad, ldy in [(0, cout), (0, 1024), (0, 1040), (16, 1040)]:
        ldx = cin + ldx_pad
        x = torch.randn(n * ih * ih, ldx, device=dev)
        y = torch.empty(n * ih * ih, ldy, device=dev)
        t_hot = timeit(lambda: ops.conv_fwd(x, ldx, 0, wt, None, None, y, ldy, 0, n, ih, ih, cin, cout, k, k, 1, True))
        def cold():
            big.add_(1.0)
            ops.conv_fwd(x, ldx, 0, wt, None, None, y, ldy, 0, n, ih, ih, cin, cout, k, k, 1, True)
        t_scrub = timeit(lambda: big.add_(1.0))
        t_cold = timeit(cold) - t_scrub
        print("cin=%4d cout=%3d k=%d ldx=%4d ldy=%4d  hot %6.1f us  cold %6.1f us" % (cin, cout, k, ldx, ldy, t_hot, t_cold))
