import sys, torch
sys.path.insert(0, ".")
from cap2det_amd import hip_ops as ops
dev = "cuda:0"
which = sys.argv[1]; n, ih, cin, cout, k, s = [int(v) for v in sys.argv[2:8]]
oh = -(-ih // s)
x = torch.randn(n * ih * ih, cin, device=dev); w = torch.randn(k * k, cin, cout, device=dev) * 0.05
wt = torch.empty(k * k, cout, cin, device=dev); ops.transpose_taps(w, wt, k * k, cin, cout)
y = torch.empty(n * oh * oh, cout, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
for _ in range(3):
    if which == "fwd": ops.conv_fwd(x, cin, 0, wt, None, None, y, cout, 0, n, ih, ih, cin, cout, k, k, s, True)
    if which == "dgrad": ops.conv_dgrad(dy, cout, 0, w, dx, cin, 0, n, ih, ih, cin, cout, k, k, s, False)
    if which == "wgrad": ops.conv_wgrad(x, cin, 0, dy, cout, 0, dw, n, ih, ih, cin, cout, k, k, s)
torch.cuda.synchronize()
