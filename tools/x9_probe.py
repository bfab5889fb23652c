"""Upper bound of a balanced pixel-major f32x9 launch: a plain row-major GEMM of the same size
(M = 2000 x 16 rows, N = 256, K = 224 x 6.25 taps) per forced tile / ring."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
  import torch
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from cap2det_amd import hip_ops as ops
  dev = "cuda:0"
  n, hw, cin, cout = 2000, 4, int(sys.argv[2]), int(sys.argv[3])
  x = torch.randn(n * hw * hw, cin, device=dev)
  w = torch.randn(1, cout, cin, device=dev) / cin ** 0.5
  if sys.argv[1] == "x9":
    keep = ops.x9_planes(w)
  y = torch.empty(n * hw * hw, cout, device=dev)
  sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
  run = lambda: ops.conv_fwd(x, cin, 0, w, sc, sh, y, cout, 0, n, hw, hw, cin, cout, 1, 1, 1, True)
  for _ in range(3):
    run()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(20):
    run()
  e.record()
  torch.cuda.synchronize()
  t = s.elapsed_time(e) / 20
  fl = 2.0 * n * hw * hw * cin * cout
  print("%-4s %-28s %7.1f us %6.1f TF  %s" % (sys.argv[1], os.environ.get("CFG", ""), t * 1e3, fl / t / 1e9,
                                               ";".join(ops.last_dispatch())))
  sys.exit(0)

for cin, cout in ((1408, 256), (1408, 224), (1024, 736)):
  print("K = %d, N = %d" % (cin, cout))
  subprocess.run([sys.executable, __file__, "fp32", str(cin), str(cout)])
  for nt in ("2", "4", "6", "22", "24", "42"):
    for bk, d in (("16", "2"), ("16", "3"), ("32", "2")):
      env = dict(os.environ, C2D_TUNE="x9_nt=%s,x9_bk=%s,x9_d=%s" % (nt, bk, d), CFG="nt%s bk%s d%s" % (nt, bk, d))
      subprocess.run([sys.executable, __file__, "x9", str(cin), str(cout)], env=env)
