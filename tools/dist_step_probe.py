"""Diagnostic: where a data-parallel train step spends its time on ONE GPU (RCCL world of one rank).  Variants: no reduction; staged joins
only (callbacks that reduce nothing); staged + bucketed all-reduce; one-shot all-reduce after the backward.
    python tools/dist_step_probe.py [steps 20]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.dist import GradSync
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29534")})
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.bfloat16)
model.cuda().train()
crit = {"obj_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
opt = FusedAdam(model.parameters(), lr=1e-3)
b = synthetic_batch((256,), 1234)
batch = (b["img"], None, b["x0bar"], b["x0"], None, b["obj"])
model._materialize(torch.device("cuda", 0))


class JoinOnly(GradSync):   # the staged callbacks (and the engine's per-stage stream joins) without any collective
    def reduce_range(self, lo, hi):
        return


def timed(sync, label):
    for _ in range(8):
        train_step(model, batch, crit, opt, True, "train", sync)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        train_step(model, batch, crit, opt, True, "train", sync)
    th = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("%-46s %.2f ms/step (host loop alone %.2f ms/step)" % (label, (t1 - t0) / STEPS * 1e3, (th - t0) / STEPS * 1e3))


class AsyncWork(GradSync):   # round 2's form: async_op=True, Work objects waited for at the end of the backward
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._works = []

    def reduce_range(self, lo, hi):
        for a, b in self.buckets(lo, hi):
            self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for w in self._works:
            w.wait()
        self._works = []
        self._seen = set()


class Tiny(AsyncWork):   # the same async calls on 4 floats: per-call cost without the bytes
    one = None

    def reduce_range(self, lo, hi):
        if Tiny.one is None:
            Tiny.one = torch.zeros(4, device="cuda")
        self._works.append(dist.all_reduce(Tiny.one, op=dist.ReduceOp.SUM, async_op=True))


class SyncCalls(GradSync):   # blocking-API form: no Work objects kept
    def reduce_range(self, lo, hi):
        for a, b in self.buckets(lo, hi):
            dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM)


class OwnCopy(GradSync):   # not a collective at all: a device-to-device copy of the same bytes on a second stream, joined with events
    side = None

    def reduce_range(self, lo, hi):
        if OwnCopy.side is None:
            OwnCopy.side = torch.cuda.Stream()
            OwnCopy.tmp = torch.empty_like(self.flat)
        cur = torch.cuda.current_stream()
        OwnCopy.side.wait_stream(cur)
        with torch.cuda.stream(OwnCopy.side):
            OwnCopy.tmp[lo:hi].copy_(self.flat[lo:hi], non_blocking=True)

    def finish(self):
        if OwnCopy.side is not None:
            torch.cuda.current_stream().wait_stream(OwnCopy.side)
        self._seen = set()


timed(None, "no reduction")
timed(AsyncWork(model._arena.grad, reduce_single=True).attach(model), "staged, async_op=True + Work.wait() (round 2)")
timed(Tiny(model._arena.grad, reduce_single=True).attach(model), "staged, async all_reduce of 4 floats per stage")
timed(SyncCalls(model._arena.grad, reduce_single=True).attach(model), "staged, blocking-API all_reduce per bucket")
timed(OwnCopy(model._arena.grad, reduce_single=True).attach(model), "staged, plain copy on a second stream instead")
s = JoinOnly(model._arena.grad, reduce_single=True).attach(model)
timed(s, "staged callbacks + stream joins, no collective")
s = GradSync(model._arena.grad, reduce_single=True).attach(model)
timed(s, "GradSync: staged + bucketed all-reduce (32 MiB)")
s = GradSync(model._arena.grad, bucket_bytes=1 << 30, reduce_single=True).attach(model)
timed(s, "staged + one all-reduce per stage")
model._grad_sync = None
s = GradSync(model._arena.grad, reduce_single=True)
timed(s, "GradSync: one-shot bucketed all-reduce after backward")
timed(None, "no reduction (again)")
dist.destroy_process_group()
