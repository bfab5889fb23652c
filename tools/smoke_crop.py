import ctypes, numpy as np, torch, time
from cap2det_amd import _lib
lib = _lib.load()
print("version", lib.c2d_version())
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = wf = 32; D = 576; N = 2000
feat = torch.randn(1, hf, wf, D, device=dev)
c = torch.rand(N, 2, device=dev); s = torch.rand(N, 2, device=dev) * 0.5 + 0.04
boxes = torch.cat([(c - s / 2).clamp(0, 1), (c + s / 2).clamp(0, 1)], 1).contiguous()
ind = torch.zeros(N, dtype=torch.int32, device=dev)
out = torch.empty(N, 7, 7, D, device=dev)
arg = torch.empty(N, 7, 7, D, dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    _lib.call("c2d_roi_crop_pool_fwd", p(feat), p(boxes), p(ind), p(out), p(arg), 1, hf, wf, D, N, 14, 2, 2, st)
run(); torch.cuda.synchronize()
# reference check with torch on GPU (not the oracle; just a smoke test)
crop = torch.empty(N, 14, 14, D, device=dev)
_lib.call("c2d_crop_and_resize_fwd", p(feat), p(boxes), p(ind), p(crop), 1, hf, wf, D, N, 14, st)
torch.cuda.synchronize()
ref = torch.nn.functional.max_pool2d(crop.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
print("fused vs unfused maxdiff", (ref - out).abs().max().item())
# numpy bilinear for a few boxes
f = feat[0].cpu().numpy(); bx = boxes.cpu().numpy(); cr = crop.cpu().numpy()
md = 0
for r in range(0, N, 97):
    y1, x1, y2, x2 = bx[r]
    for y in range(14):
        iy = np.float32(y1 * np.float32(hf - 1)) + np.float32(y) * np.float32((y2 - y1) * np.float32(hf - 1) / np.float32(13))
        for x in range(14):
            ix = np.float32(x1 * np.float32(wf - 1)) + np.float32(x) * np.float32((x2 - x1) * np.float32(wf - 1) / np.float32(13))
            if iy < 0 or iy > hf - 1 or ix < 0 or ix > wf - 1: v = 0 * f[0, 0]
            else:
                t, b_, l, rr = int(np.floor(iy)), int(np.ceil(iy)), int(np.floor(ix)), int(np.ceil(ix))
                ly, lx = iy - t, ix - l
                top = f[t, l] + (f[t, rr] - f[t, l]) * lx; bot = f[b_, l] + (f[b_, rr] - f[b_, l]) * lx
                v = top + (bot - top) * ly
            md = max(md, np.abs(v - cr[r, y, x]).max())
print("crop vs numpy maxdiff", md)
for fn, byts in ((run, N * 49 * D * 4),):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("fused crop+pool: %.3f ms  %.1f GB/s (algorithmic write bytes)" % (ms, byts / ms / 1e6))
dfeat = torch.zeros_like(feat); dout = torch.randn_like(out)
_lib.call("c2d_roi_crop_pool_bwd", p(dout), p(arg), p(boxes), p(ind), p(dfeat), 1, hf, wf, D, N, 14, 2, 2, st)
torch.cuda.synchronize(); print("bwd sum", dfeat.sum().item(), dout.sum().item())
