"""Per-kernel register / LDS / spill table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

  python tools/kernel_resources.py cap2det_amd/csrc/conv_gemm.hip [name filter] [extra hipcc flags ...]
"""
import re
import subprocess
import sys


def main():
  src = sys.argv[1]
  flt = sys.argv[2] if len(sys.argv) > 2 else ""
  extra = sys.argv[3:]
  cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
  out = subprocess.run(cmd, capture_output=True, text=True).stderr
  cur, rows = None, []
  for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
      cur = {"name": m.group(1)}
      rows.append(cur)
      continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
      cur[m.group(1).strip()] = int(m.group(2))
  demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                            capture_output=True, text=True).stdout.splitlines()
  for r, d in zip(rows, demangle):
    d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    if flt and flt not in d:
      continue
    print("%-70s vgpr %3d agpr %3d spill %3d sgpr %3d occ %d lds %6d scratch %d" % (
        d[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("TotalSGPRs", -1),
        r.get("Occupancy", -1), r.get("LDS Size", -1), r.get("ScratchSize", -1)))


if __name__ == "__main__":
  main()
