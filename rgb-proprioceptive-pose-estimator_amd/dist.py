"""Data-parallel replication: one process per GPU, parameters replicated, per-step SUM all-reduce of the
flat gradient buffer over RCCL (torch.distributed backend "nccl" on ROCm) or gloo on CPU.

Replaces the single-process nn.DataParallel wrappers of the reference (models/naive.py:224,234,253,274).
The loss is a SUM over samples (models/losses.py:75,80,118,122), so the gradient of a global batch is
the SUM of shard gradients: the reduction op is SUM, never AVG.  BatchNorm statistics stay per replica,
as they do under nn.DataParallel.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("RPE_DIST_FORCE_INIT")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # RPE_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal (RCCL wants one GPU per rank)
            backend = os.environ.get("RPE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, world, local


def shard_bounds(n, rank, world):
    """Episodes [lo, hi) of rank `rank` (shard along N, never along the LSTM time axis S)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradSync:
    """Bucketed SUM all-reduce over a flat gradient buffer.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of M bytes moves 2*(P-1)/P*M
    through one link, so a few large buckets (default 32 MiB) keep per-call latency negligible while letting
    the first buckets start before the last ones are queued.

    Two ways to use it:
      * all_reduce(): one shot after backward (what the reference's DataParallel gather amounts to);
      * attach(model): the model's backward then calls stage_done(name) as the gradients of "fc" (trunk fc + all head
        layers), "layer4" .. "layer1", "stem" become final, and each stage's contiguous arena slice is reduced
        asynchronously while the earlier layers are still being differentiated; finish() waits for all of them.
    """

    def __init__(self, flat_grad, bucket_bytes=32 << 20, group=None, reduce_single=False):
        self.flat = flat_grad
        self.group = group
        self.reduce_single = reduce_single   # issue the collectives even in a 1-rank group (RCCL bring-up test on one GPU)
        self.bucket = max(1, bucket_bytes // flat_grad.element_size())
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._slices = None
        self._seen = None
        # Device tensors: the collectives are issued through c10d's BLOCKING-API form from a dedicated stream ("comm") that first
        # waits for the compute stream; finish() makes the compute stream wait for it.  That overlaps them with the rest of the
        # backward exactly as async_op=True would -- which is what round 2 used and what measured +9.5 ms PER STEP on one MI355X
        # (RCCL world of one rank, any number of calls, even on 4 floats: tools/dist_step_probe.py, profiles/r03_dist_step_probe.txt):
        # with async Work objects outstanding the host's launch loop slows from 12.5 to 18.3 ms per step and the device starves;
        # the blocking-API form costs nothing measurable (20.35 vs 20.30 ms).
        self._comm = None
        self._pending = False
        # With blocking waits every collective would stall the HOST inside the backward's issue loop (the stage boundaries sit between
        # the trunk engine's launches), serialising launch and exchange: refuse loudly rather than run 1.5x slower silently.
        for var in ("TORCH_NCCL_BLOCKING_WAIT", "NCCL_BLOCKING_WAIT"):
            if os.environ.get(var, "0") not in ("0", ""):
                import warnings
                warnings.warn("%s=%s: the staged gradient exchange issues its collectives inside the backward and relies on them being "
                              "asynchronous to the host; with blocking waits each stage boundary stalls the launch loop" % (var, os.environ[var]))

    def buckets(self, lo=0, hi=None):
        hi = self.flat.numel() if hi is None else hi
        return [(a, min(hi, a + self.bucket)) for a in range(lo, hi, self.bucket)]

    def reduce_range(self, lo, hi):
        if (self.world == 1 and not self.reduce_single) or hi <= lo:
            return
        if not self.flat.is_cuda:   # gloo on host tensors (tests): nothing to overlap with
            for a, b in self.buckets(lo, hi):
                dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
            return
        cur = torch.cuda.current_stream(self.flat.device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(self.flat.device)
        self._comm.wait_stream(cur)          # the gradients of this range are final at this point of the compute stream
        with torch.cuda.stream(self._comm):
            for a, b in self.buckets(lo, hi):
                dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
        self._pending = True

    def defer_add(self, carry):
        """Gradient accumulation across backward() calls (torch semantics: .grad accumulates until zero_grad()): `carry` -- the flat
        gradient as it stood before this backward, already reduced -- is added to the buffer once this backward's staged reduction has
        finished (finish()); adding it earlier would race with the collectives in flight."""
        self._carry = carry if getattr(self, "_carry", None) is None else self._carry + carry

    def finish(self):
        if self._pending:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._comm)
            self._pending = False
        if getattr(self, "_carry", None) is not None:
            self.flat.add_(self._carry)
            self._carry = None
        if self._slices is not None and self._seen is not None:
            if (self.world > 1 or self.reduce_single) and self._seen != set(self._slices):
                raise RuntimeError("GradSync: stages %s were never reduced" % sorted(set(self._slices) - self._seen))
            self._seen = set()

    def all_reduce(self):
        self.reduce_range(0, self.flat.numel())
        self.finish()

    # -- staged mode --------------------------------------------------------------------------------
    def attach(self, model):
        """Derive the arena slice of every backward stage of `model` and register with it."""
        self._slices = stage_slices(model)
        self._seen = set()
        model._grad_sync = self
        return self

    @property
    def staged(self):
        return self._slices is not None

    def stage_done(self, name):
        lo, hi = self._slices[name]
        self._seen.add(name)
        self.reduce_range(lo, hi)


def stage_slices(model):
    """name -> [lo, hi) element range of the flat gradient arena, in backward completion order.

    The arena lays parameters out in registration order: trunk (conv1/bn1, layer1..layer4, fc) first, then the heads, so
    every stage is one contiguous range: "fc" = trunk fc + every head layer (their gradients exist before the trunk's
    backward starts), then layer4, layer3, layer2, and finally layer1 + the stem."""
    arena = model._arena
    if arena is None:
        raise RuntimeError("stage_slices: the model has no parameter arena yet (run one forward on the device)")
    trunk = model.trunk
    off = {id(p): o for p, o in zip(arena.params, arena.offsets)}

    def first(mod):
        return min(off[id(p)] for p in mod.parameters())

    marks = [("stem", 0), ("layer2", first(trunk.layer2)), ("layer3", first(trunk.layer3)), ("layer4", first(trunk.layer4)),
             ("fc", first(trunk.fc)), ("end", arena.numel)]
    if first(trunk.conv1) != 0 or [m[1] for m in marks] != sorted(m[1] for m in marks):
        raise RuntimeError("stage_slices: trunk parameters are not laid out first and in order in the arena")
    s = {"stem": (0, marks[1][1]), "layer2": (marks[1][1], marks[2][1]), "layer3": (marks[2][1], marks[3][1]),
         "layer4": (marks[3][1], marks[4][1]), "fc": (marks[4][1], marks[5][1])}
    # layer1 finishes before the stem but shares its slice: reduce the slice once, when the stem is done
    s["layer1"] = (0, 0)
    # A frozen trunk body (feature_extract and use_pretrained, util/model_utils.py:110-113 of the reference: every published job) leaves
    # ~23.5 M gradient elements that are zero on every rank: exchanging them (94 MB per step under a third of the full step's work)
    # buys nothing.  Every stage is clipped to the span of the arena's TRAINABLE segments inside it; a stage without any becomes empty.
    segs = arena.trainable_segments()
    for name, (lo, hi) in list(s.items()):
        inside = [(max(lo, a), min(hi, b)) for a, b in segs if min(hi, b) > max(lo, a)]
        s[name] = (min(a for a, _ in inside), max(b for _, b in inside)) if inside else (lo, lo)
    return s


def broadcast_parameters(flat_params, buffers=(), src=0, group=None):
    """Make every replica start from rank `src`'s parameters and BN buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat_params, src, group=group)
    for b in buffers:
        dist.broadcast(b, src, group=group)
