"""Counting the GPUs of this box without loading torch or HIP (the parent process of the
one-process-per-GPU ranks — `python bench.py --gpus N`, the analogue of the reference's
train_wsod.sh:46-88 launcher — must never initialise a device it then hands to its children)."""
import os


def count_visible_gpus():
  """Number of GPUs this process tree may use, WITHOUT touching HIP (the parent of the rank
  processes must stay GPU-free): KFD topology nodes with SIMDs (`simd_count > 0`; CPU nodes report
  0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists."""
  override = os.environ.get("C2D_NUM_GPUS")
  if override is not None:                     # explicit answer of the operator: trusted as is
    return max(0, int(override))
  root = "/sys/class/kfd/kfd/topology/nodes"
  n = 0
  try:
    nodes = sorted(os.listdir(root))
  except OSError:
    nodes = None
  if nodes is None:
    # /dev/kfd passed into a container without the KFD sysfs tree: one render node per GPU
    # (still no HIP call: the parent stays GPU-free)
    try:
      n = len([e for e in os.listdir("/dev/dri") if e.startswith("renderD")])
    except OSError:
      n = 0
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
      # an index outside [0, n) ends the list (the HIP runtime ignores everything behind it);
      # UUID entries (GPU-xxxx) cannot be checked without the runtime and count as given
      listed = []
      for t in (t.strip() for t in v.split(",")):
        if t == "":
          continue
        if t.lstrip("-").isdigit() and not 0 <= int(t) < n:
          break
        listed.append(t)
      n = min(n, len(listed))
  return n
