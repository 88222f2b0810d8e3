"""RCCL readiness on the hardware a 1-GPU box has: ONE fresh child process joins a
world-size-1 process group with backend "nccl" (= RCCL on ROCm) and

  * all-reduces the real FFNO2D gradient bucket on the device through FlatGradBucket.all_reduce_mean() -- proves that
    librccl loads, that the environment defaults of rpde/launch.py (HSA_ENABLE_IPC_MODE_LEGACY=0) do not break
    communicator init, and that the collective runs on the launch stream (the optimizer step that follows sees it);
  * captures the whole training step INCLUDING that all-reduce as one hipGraph and replays it (round 4: what a rank's host
    does per step shrinks from ~250 launches to one);
  * runs one training epoch + evaluation through rpde/entry.py (main_2d's body) with every collective enabled.

What this does NOT show: any N > 1 behaviour (xGMI transport, ring / tree selection, scaling).  That stays
unmeasured until the driver's 8-GPU run.  Reference counterpart: main_2d.py:89-94,147-149.
"""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import DROPIN, REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
sys.path[:0] = [os.environ["RPDE_T_REPO"], os.environ["RPDE_T_DROPIN"]]
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from models.ffno import FFNO2D
from rpde.optim import FlatAdamW
from rpde.parallel import FlatGradBucket
from utils.loss import RelativeL2Loss
from utils.synthetic import advance, random_fields
cfg = dict(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True,
           n_ff_layers=3, layer_norm=True, dropout=0.1)
torch.manual_seed(0)
model = FFNO2D(**cfg).to("cuda:0").train()
bucket = FlatGradBucket(model.parameters())
opt = FlatAdamW(model.parameters(), lr=1e-3, bucket=bucket, capturable=True)
x = random_fields(2, 64, 2, seed=3); y = advance(x, 2)
x, y = x.cuda(), y.cuda()
bucket.zero()
loss = RelativeL2Loss()(model(x), y)
loss.backward()
bucket.gather()
before = bucket.flat.clone()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
bucket.all_reduce_mean()            # RPDE_FORCE_DIST=1: the collective runs although world == 1
ev1.record()
opt.step()
torch.cuda.synchronize()
same = bool(torch.equal(before, bucket.flat))
print(json.dumps({"rccl": True, "backend": dist.get_backend(), "bucket_bytes": bucket.nbytes, "sum_of_one_rank_is_identity": same,
                  "allreduce_ms_first_call": round(ev0.elapsed_time(ev1), 3), "loss": float(loss.detach())}), flush=True)
# the whole step -- forward, loss, backward, the RCCL all-reduce, the optimizer -- captured as ONE hipGraph and replayed
# (rpde/graph.py; dropout masks advance through the device-side counter)
from rpde.graph import GraphedTrainStep
del loss
gs = GraphedTrainStep(model, RelativeL2Loss(), opt, x, y, warmup=1, after_backward=bucket.all_reduce_mean)
gl = [float(gs(x, y)) for _ in range(6)]
torch.cuda.synchronize()
print(json.dumps({"graphed": True, "losses": gl, "finite": all(bool(torch.isfinite(p_).all()) for p_ in model.parameters())}), flush=True)
# main_2d's body (reuses the process group and tears it down at its end), every collective on (gradient all-reduce per step, loss / metric reductions)
from rpde.entry import run
l2 = run(2, ["model=ffno_2d/ffno_2d", "dataset=synthetic/ns_mres", "dataset.resolutions={32: 16}", "dataset.n_val=8",
             "dataset.n_test=8", "model.width=64", "model.n_layers=2", "model.n_modes=8", "training.epochs=1",
             "training.batch_size=8", "checkpoint_dir=" + os.environ["RPDE_T_TMP"]])
print(json.dumps({"entry_test_rel_l2": l2}), flush=True)
'''


def test_single_rank_nccl_process_group_runs_the_gradient_bucket_and_the_entry_point(gpu_device, tmp_path):
    from rpde.launch import free_port, rank_env
    env = rank_env(0, 1, free_port())
    env.update(RPDE_FORCE_DIST="1", RPDE_T_REPO=REPO, RPDE_T_DROPIN=DROPIN, RPDE_T_TMP=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    first = [d for d in recs if d.get("rccl")]
    assert first and first[0]["backend"] == "nccl" and first[0]["sum_of_one_rank_is_identity"], recs
    assert first[0]["bucket_bytes"] == 6828576
    graphed = [d for d in recs if d.get("graphed")]
    assert graphed and graphed[0]["finite"] and len(set(graphed[0]["losses"])) > 3 and graphed[0]["losses"][-1] < graphed[0]["losses"][0], recs
    last = [d for d in recs if "entry_test_rel_l2" in d]
    assert last and 0 < last[0]["entry_test_rel_l2"] < 2.0, recs
    print("\n[rccl]", first[0])
