"""Timings of the §8f rows on one MI355X: multi-class NMS at the BASELINE sizes, multi-scale
inference of one image, and the input pipeline (host JPEG decode per core, GPU preprocessing)."""
import io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from cap2det_amd import hip_ops as ops
from cap2det_amd.readers import tfrecord as T
from tests import util_model
dev = "cuda:0"

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

rng = np.random.default_rng(0)
for n, c in ((2000, 20), (2000, 80)):
    boxes = torch.from_numpy(util_model.synthetic_boxes(rng, n)[None]).to(dev)
    scores = torch.softmax(torch.randn(1, n, c + 1, device=dev) * 3, -1)[..., 1:].contiguous()
    ws = torch.empty(ops._lib.load().c2d_multiclass_nms_workspace_bytes(1, n, c, 100), dtype=torch.uint8, device=dev)
    t = timeit(lambda: ops.multiclass_nms(boxes, scores, c, 0, c, 1e-5, 0.3, 100, 300, ws))
    num = ops.multiclass_nms(boxes, scores, c, 0, c, 1e-5, 0.3, 100, 300, ws)[0]
    print("multiclass_nms N=%d C=%d: %.3f ms (num_detections %d)" % (n, c, t, int(num[0])))

# multi-scale inference (eval_min_dimension 1200/800/600/400) of one 375x500 image, N=2000
from cap2det_amd.models import builder
pl = util_model.load_pipeline()
model = builder.build(pl.model, is_training=False, device=dev)
classes = model.label_extractor.classes
ex = util_model.make_examples(rng, 1, 375, 500, 2000, [2000], classes)
d = dict(ex)
for k in ("image", "proposals"): d[k] = torch.from_numpy(ex[k]).to(dev)
d["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(dev)
t = timeit(lambda: model.build_prediction(d), iters=5, warm=2)
print("multi-scale inference (4 scales + NMS x4), 375x500, N=2000: %.1f ms / image" % t)
t = timeit(lambda: model.build_prediction(d, single_scale=True), iters=5, warm=2)
print("single-scale inference + NMS x4, 375x500: %.1f ms / image" % t)

# input pipeline
from PIL import Image
y, x = np.mgrid[0:375, 0:500]
img = np.clip(np.stack([128 + 90 * np.sin(x / 19.0 + c) * np.cos(y / 15.0 - c) for c in range(3)], -1) + rng.normal(0, 8, (375, 500, 3)), 0, 255).astype(np.uint8)
b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=92); data = b.getvalue()
t0 = time.perf_counter()
for _ in range(50): out = T.decode_jpeg(data)
t1 = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
for _ in range(50): np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
t2 = (time.perf_counter() - t0) / 50
print("JPEG decode 375x500 (%d KB): native %.2f ms, libjpeg-turbo/Pillow %.2f ms per image per core" % (len(data) // 1024, t1 * 1e3, t2 * 1e3))
u8 = torch.from_numpy(out).to(dev)
canvas = torch.empty(1000, 1333, 3, device=dev)
t = timeit(lambda: ops.image_resize_pad_u8(u8, True, canvas, 1000, 1333))
print("GPU flip + resize 375x500 -> 1000x1333 fp32 + pad: %.3f ms (%.1f GB/s written)" % (t, canvas.numel() * 4 / t / 1e6))
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as pool:
    t0 = time.perf_counter(); list(pool.map(T.decode_jpeg, [data] * 320)); t3 = time.perf_counter() - t0
print("JPEG decode, 16 host threads: %.0f images/s" % (320 / t3))
