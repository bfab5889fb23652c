"""Counting the GPUs of this box without loading torch or HIP (the parent process of the
one-process-per-GPU ranks — `python bench.py --gpus N`, the analogue of the reference's
train_wsod.sh:46-88 launcher — must never initialise a device it then hands to its children)."""
import os


def count_visible_gpus():
  """Number of GPUs this process tree may use, WITHOUT touching HIP (the parent of the rank
  processes must stay GPU-free): KFD topology nodes with SIMDs (`simd_count > 0`; CPU nodes report
  0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists."""
  root = "/sys/class/kfd/kfd/topology/nodes"
  n = 0
  try:
    nodes = sorted(os.listdir(root))
  except OSError:
    nodes = []
  for node in nodes:
    try:
      with open(os.path.join(root, node, "properties")) as f:
        props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
      if int(props.get("simd_count", "0")) > 0:
        n += 1
    except (OSError, ValueError):
      continue
  for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    v = os.environ.get(var)
    if v is not None:
      listed = [t for t in v.split(",") if t.strip() != ""]
      n = min(n, len(listed))
  return n
