"""Interleaved A/B sweep of the igemm tile configurations / row orders on the second-stage conv
shapes (N = 2000 ROIs).  Needs C2D_TUNE=1 in the environment BEFORE the library is loaded:

  C2D_TUNE=1 python tools/sweep_igemm.py [fwd|dgrad]
"""
import os, sys, torch
os.environ.setdefault("C2D_TUNE", "1")
sys.path.insert(0, ".")
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
SHAPES = [  # n, ih, cin, cout, k, stride   (Mixed_5a/5b/5c)
    (2000, 7, 576, 128, 1, 1), (2000, 7, 576, 192, 1, 1), (2000, 7, 128, 192, 3, 2),
    (2000, 7, 192, 256, 3, 1), (2000, 7, 256, 256, 3, 2),
    (2000, 4, 1024, 352, 1, 1), (2000, 4, 1024, 192, 1, 1), (2000, 4, 1024, 160, 1, 1),
    (2000, 4, 1024, 128, 1, 1), (2000, 4, 192, 320, 3, 1), (2000, 4, 160, 224, 3, 1),
    (2000, 4, 192, 224, 3, 1), (2000, 4, 224, 224, 3, 1)]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
VARIANTS = [("rm64", "1", "2", "0"), ("rm128", "1", "3", "0"), ("pm64", "0", "2", "0"), ("pm128", "0", "3", "0"),
            ("SKrm64", "1", "2", "1"), ("SKrm128", "1", "3", "1"), ("SKpm64", "0", "2", "1"), ("SKpm128", "0", "3", "1")]
ws = torch.zeros(ops.conv_workspace_bytes(), dtype=torch.uint8, device=dev)
ops.set_conv_workspace(ws)
def run(fn, iters):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (n, ih, cin, cout, k, st) in SHAPES:
    oh = -(-ih // st)
    x = torch.randn(n * ih * ih, cin, device=dev)
    w = torch.randn(k * k, cin, cout, device=dev) / (k * k * cin) ** 0.5
    wt = torch.empty(k * k, cout, cin, device=dev); ops.transpose_taps(w, wt, k * k, cin, cout)
    y = torch.empty(n * oh * oh, cout, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    fl = 2.0 * n * oh * oh * cin * cout * k * k
    if which == "fwd":
        fn = lambda: ops.conv_fwd(x, cin, 0, wt, sc, sh, y, cout, 0, n, ih, ih, cin, cout, k, k, st, True)
    else:
        fn = lambda: ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, st, False)
    best = {}
    for rnd in range(4):
        for name, rm, cfg, sk in VARIANTS:
            if k == 1 and "pm" in name: continue
            os.environ["C2D_IGEMM_ROW_MAJOR"] = rm; os.environ["C2D_IGEMM_CFG"] = cfg
            os.environ["C2D_IGEMM_SK"] = sk
            if rnd == 0: run(fn, 2)
            t = run(fn, 10)
            best[name] = min(best.get(name, 1e9), t)
    print("%s n=%4d %dx%d cin=%4d cout=%3d k=%d s=%d | " % (which, n, ih, ih, cin, cout, k, st) +
          " ".join("%s %5.1f" % (nm, t * 1e3) for nm, t in best.items()) + " | best %s %.1f TF" % (min(best, key=best.get), fl / min(best.values()) / 1e9))
