"""Rehearsal of the RCCL gradient exchange on ONE GPU (run as a child process by
tests/test_gpu_rccl.py; also `python tools/rccl_rehearsal.py` on a GPU box).

The reference's only parallel mode is one process per GPU (/root/reference/train_wsod.sh:46-88);
here that is `torch.distributed` over RCCL with one all-reduce of the flat gradient bucket per step
(cap2det_amd/train/data_parallel.py).  A pool box has one GPU, so the code that a multi-GPU run
executes — `init_process_group("nccl")`, the asynchronous all-reduce on RCCL's stream beside the
step's compute / filter-gradient / look-ahead streams, the stream joins around it, the collective
between the two hipGraph replays — is exercised here at world size 1 with C2D_FORCE_ALLREDUCE=1
(a one-rank sum is the identity): three full-size eager steps and three hipGraph steps with the
collectives must reproduce the same steps without them — bitwise in the first forward pass, to
the order of the filter gradients' fp32 atomics in the first update.

Prints one JSON line; exit code 0 = all comparisons hold."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def main():
  import numpy as np
  import torch
  import torch.distributed as dist
  from cap2det_amd import synthetic
  from cap2det_amd.train import data_parallel
  from cap2det_amd.train.trainer import Trainer

  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  if "MASTER_PORT" not in os.environ:
    with socket.socket() as sock:
      sock.bind(("127.0.0.1", 0))
      os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
  torch.cuda.set_device(0)
  dist.init_process_group(backend="nccl", rank=0, world_size=1)
  dev = "cuda:0"
  size = os.environ.get("C2D_REHEARSAL_SIZE", "full")
  hw, n = (500, 2000) if size == "full" else (96, 64)

  calls = {"n": 0}
  real_all_reduce = dist.all_reduce
  def counting_all_reduce(*a, **k):
    calls["n"] += 1
    return real_all_reduce(*a, **k)
  dist.all_reduce = counting_all_reduce

  pipeline = synthetic.load_pipeline()
  batches = None

  def run(force, use_graph=False):
    nonlocal batches
    if force:
      os.environ["C2D_FORCE_ALLREDUCE"] = "1"
    else:
      os.environ.pop("C2D_FORCE_ALLREDUCE", None)
    assert data_parallel.collectives_on() == bool(force)
    before = calls["n"]
    trainer = Trainer(pipeline, device=dev, seed=21)
    classes = trainer.model.label_extractor.classes
    if batches is None:
      rng = np.random.default_rng(5)
      batches = []
      for _ in range(2):
        ex = synthetic.make_examples(rng, 1, hw, hw, n, [n], classes)
        d = dict(ex)
        for k in ("image", "proposals", "number_of_proposals"):
          d[k] = torch.from_numpy(ex[k]).to(dev).contiguous()
        batches.append(d)
    out = dict(losses=[], scores0=None, after_first=None)
    for i in range(3):
      losses = trainer.train_step(batches[i % 2], dropout_seed=40 + i,
                                  prefetch=None if use_graph else batches[(i + 1) % 2])
      torch.cuda.synchronize()
      out["losses"].append({k: float(v.item()) for k, v in losses.items()})
      if i == 0:
        out["scores0"] = [trainer.predictions["oicr_proposal_scores_at_%d" % j].detach().clone()
                          for j in range(4)]
        lo, hi = trainer.bucket
        out["after_first"] = trainer.model.store.values[lo:hi].clone()
    out["collectives"] = calls["n"] - before
    out["streams"] = dict(side=trainer.model.engine.second.side is not None,
                          lookahead=trainer.model.engine.prefetch_stream is not None)
    return out

  report = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "size": size,
            "checks": []}
  ok = True
  for use_graph in (False,):
    ref = run(False, use_graph)
    got = run(True, use_graph)
    name = "graph" if use_graph else "eager"
    chk = {"mode": name, "collectives_without": ref["collectives"], "collectives_with": got["collectives"]}
    # without the switch a one-rank group issues nothing; with it: four per eager step (one per
    # second-stage block as the backward pass leaves it — Mixed_5c with the heads, 5b, 5a: issued
    # from the filter-gradient stream, data_parallel.BlockReducer — then the Mixed_4e prefix; two
    # with C2D_DP_BUCKETS=two), one per graph step (between the two replays)
    per_step = 2 if os.environ.get("C2D_DP_BUCKETS") == "two" else 4
    good = ref["collectives"] == 0 and got["collectives"] == (3 if use_graph else 3 * per_step)
    if not use_graph:
      good = good and got["streams"]["side"] and got["streams"]["lookahead"]
    fwd_equal = all(torch.equal(a, b) for a, b in zip(ref["scores0"], got["scores0"]))
    chk["first_forward_bitwise_equal"] = fwd_equal
    good = good and fwd_equal
    worst = 0.0
    for k, v in ref["losses"][0].items():
      worst = max(worst, abs(v - got["losses"][0][k]) / max(abs(v), 1e-12))
    chk["first_step_loss_rel_diff"] = worst
    good = good and worst <= 2e-6
    a, b = ref["after_first"].double(), got["after_first"].double()
    upd = float((a - b).abs().max() / a.abs().max())
    chk["first_update_max_diff_of_scale"] = upd
    good = good and upd <= 5e-5
    finite = all(np.isfinite(v) for step in got["losses"] for v in step.values())
    chk["all_losses_finite"] = finite
    good = good and finite
    chk["ok"] = bool(good)
    ok = ok and good
    report["checks"].append(chk)
  # The per-block exchange (the RCCL default: asynchronous all-reduces issued from INSIDE the
  # filter-gradient stream, joined with Work.wait() on the main stream, the side stream itself
  # joined at the end of engine.backward) against the two-bucket form, both forced at one rank.
  # A collective that started before its block's last writer, or an optimiser that ran before a
  # collective finished, loses a whole layer's contribution — an O(1) error; what two correct runs
  # differ by is the order of the remaining fp32 atomics (heads / Mixed_4e filter gradients, loss
  # scalars and split-K filter gradients: ~1e-6 of the bucket's scale).  So: the two forms must
  # agree to 1e-5 of scale on the reduced gradients and on the updated variables, and no worse
  # than ten times what two runs of the SAME form differ by.
  def grads_after_first_step(buckets):
    os.environ["C2D_FORCE_ALLREDUCE"] = "1"
    os.environ["C2D_DP_BUCKETS"] = buckets
    try:
      trainer = Trainer(pipeline, device=dev, seed=21)
      before = calls["n"]
      trainer.train_step(batches[0], dropout_seed=40, prefetch=batches[1])
      torch.cuda.synchronize()
      lo, hi = trainer.bucket
      return trainer.model.store.grads[lo:hi].clone(), trainer.model.store.values[lo:hi].clone(), calls["n"] - before
    finally:
      for k in ("C2D_DP_BUCKETS", "C2D_FORCE_ALLREDUCE"):
        os.environ.pop(k, None)
  g_blocks, v_blocks, n_blocks = grads_after_first_step("blocks")
  g_again, v_again, _ = grads_after_first_step("blocks")
  g_two, v_two, n_two = grads_after_first_step("two")
  def rel(a, b):
    return float((a.double() - b.double()).abs().max() / a.double().abs().max())
  chk = {"mode": "blocks_vs_two", "collectives_blocks": n_blocks, "collectives_two": n_two,
         "same_form_gradient_diff_of_scale": rel(g_blocks, g_again),
         "gradient_diff_of_scale": rel(g_blocks, g_two),
         "updated_variables_diff_of_scale": rel(v_blocks, v_two)}
  bound = max(1e-5, 10.0 * chk["same_form_gradient_diff_of_scale"])
  chk["ok"] = bool(chk["gradient_diff_of_scale"] <= bound and
                   chk["updated_variables_diff_of_scale"] <= 1e-5 and n_blocks == 4 and n_two == 2)
  ok = ok and chk["ok"]
  report["checks"].append(chk)
  # what the process group itself counts
  ones = torch.ones(1, device=dev, dtype=torch.int32)
  real_all_reduce(ones)
  report["ranks_counted_by_all_reduce"] = int(ones.item())
  report["ok"] = bool(ok and report["ranks_counted_by_all_reduce"] == 1)
  print(json.dumps(report))
  sys.stdout.flush()
  dist.destroy_process_group()
  return 0 if report["ok"] else 1


if __name__ == "__main__":
  sys.exit(main())
