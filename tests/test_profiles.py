"""Guards the committed measurement evidence: every profiles/*.json parses,
the newest PMC summaries bench.py quotes carry kernel families, every file profiles/README.md
names exists and is non-empty, and the summarisers write atomically."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def test_every_profile_json_parses():
  files = sorted(glob.glob(os.path.join(PROFILES, "*.json")))
  assert len(files) > 20
  for f in files:
    assert os.path.getsize(f) > 2, f
    with open(f) as fh:
      json.load(fh)


@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_newest_pmc_summaries_have_families(cfg):
  for kind, keys in (("traffic", ("hbm_bytes_per_step", "hbm_bytes_per_launch")),
                     ("mfma", ("mfma_busy",))):
    newest = sorted(glob.glob(os.path.join(PROFILES, "r*_%s_%s.json" % (kind, cfg))))[-1]
    with open(newest) as fh:
      fams = json.load(fh)["families"]
    gemm = [k for k in fams if k.startswith("igemm") or k.startswith("wgrad")]
    assert gemm, (newest, sorted(fams))
    for k in gemm:
      for key in keys:
        assert key in fams[k], (newest, k, key)


def test_readme_names_existing_files():
  with open(os.path.join(PROFILES, "README.md")) as fh:
    text = fh.read()
  names = set(re.findall(r"`(r\d\d_[A-Za-z0-9_.{},]+?\.(?:json|csv|txt))`", text))
  assert names
  checked = 0
  def expand(name):          # shell brace expansion: r04_x_c{1,2}{,_serial}.csv
    m = re.search(r"\{([^}]*)\}", name)
    if not m:
      return [name]
    return [e for v in m.group(1).split(",") for e in expand(name[:m.start()] + v + name[m.end():])]

  for name in names:
    for v in expand(name):
      path = os.path.join(PROFILES, v)
      assert os.path.isfile(path), "profiles/README.md names a missing file: %s" % v
      assert os.path.getsize(path) > 2, "profiles/README.md names an empty file: %s" % v
      checked += 1
  assert checked >= 10


def test_atomic_writer_leaves_no_partial_file(tmp_path):
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  try:
    from _atomic import write_json
  finally:
    sys.path.pop(0)
  target = tmp_path / "out.json"
  write_json(str(target), {"families": {"a": 1}}, indent=1, sort_keys=True)
  assert json.loads(target.read_text()) == {"families": {"a": 1}}
  # a document json cannot encode must leave the previous content untouched (round 4: `{\n`)
  with pytest.raises(TypeError):
    write_json(str(target), {"families": {None: 1, "b": 2}}, indent=1, sort_keys=True)
  assert json.loads(target.read_text()) == {"families": {"a": 1}}
  assert [p.name for p in tmp_path.iterdir()] == ["out.json"]


def test_bench_reports_unparsable_summaries(tmp_path, monkeypatch):
  """bench.py's PMC lookup: newest file that parses wins, the broken one is named in the line."""
  src = open(os.path.join(ROOT, "bench.py")).read()
  assert "pmc_summaries_error" in src
  assert "except Exception:\n        pass" not in src


def test_every_gemm_kernel_of_the_committed_profiles_has_a_pmc_family():
  """tools/summarize_pmc.family_of decides which roofline family a kernel's counters go to; a GEMM
  kernel it does not know silently drops out of `traffic` / `mfma_busy` (round 5 found the grouped
  bf16 1x1 filter-gradient kernel unclassified).  Every igemm / wgrad / roi / pool / bn / adagrad
  kernel name of the newest committed kernel-stats summaries must map to a family, bf16 instances to
  the bf16 families."""
  import csv
  import sys
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  from summarize_pmc import family_of
  seen = 0
  for cfg in ("c1", "c2"):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats_%s_serial.csv" % cfg)))
    assert files, cfg
    with open(files[-1]) as f:
      for row in csv.DictReader(f):
        name = row["Name"]
        if not re.search(r"igemm|wgrad|roi_crop|roi_bwd|pool3x3|bn_relu_bwd|adagrad", name):
          continue
        fam = family_of(name)
        assert fam, (files[-1], name)
        seen += 1
        if re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 2, (?:true|false)(?:, 1)?>", name):
          assert fam == "igemm_bf16", (name, fam)
        if "bf16" in name and "wgrad" in name:
          assert fam == "wgrad_bf16", (name, fam)
        # fp32 operands as nine bf16 partial products: families of their own (priced against the
        # bf16 pipe / 9, never against the fp32 matrix peak)
        if re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 4, (?:true|false), 3>", name):
          assert fam == "igemm_x9", (name, fam)
        if "wgrad1x1_x9_kernel" in name or "wgrad3x3_x9_kernel" in name:
          assert fam == "wgrad_x9", (name, fam)
  assert seen >= 30
  from summarize_mfma import family as mfma_family
  assert mfma_family("void c2d_ig::(anonymous namespace)::wgrad3x3_x9_kernel<2, true>(Wgrad3X9Args)") == "wgrad_x9"
  assert mfma_family("void c2d_ig::igemm_ring_kernel<0, 4, 1, 1, 3, true, 32, 2, 4, false, 3>(IgemmArgs)") == "igemm_x9"
