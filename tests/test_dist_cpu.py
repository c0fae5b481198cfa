"""Multi-process (gloo, world_size 2) checks of the data-parallel path on CPU: episode sharding, the bucketed SUM
all-reduce over a flat gradient buffer, and the 'SUM, not AVG' property of a summed loss.

What runs here is dist.py (GradSync, shard_bounds, broadcast_parameters) around a STAND-IN regressor with the path's loss structure:
the pose models themselves have no CPU path (the HIP extension is the product and refuses CPU tensors), so the model-level
data-parallel checks live in tests/test_gpu_dist.py (two gloo ranks sharing one GPU: staged == one-shot reduction; RCCL with one
rank) and, for the staged joins' cost on one GPU, `bench.py --force-dist`."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rgb_proprioceptive_pose_estimator_amd.dist import GradSync, broadcast_parameters, init_from_env, shard_bounds, stage_slices
from rgb_proprioceptive_pose_estimator_amd.params import ParamArena


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # a stand-in regressor with the path's loss structure: SUM over samples of a per-sample distance
    W = torch.randn(7, 5, requires_grad=True)
    x = torch.randn(6, 5)
    t = torch.randn(6, 7)

    def loss_of(rows):
        return torch.sqrt(((x[rows] @ W.t() - t[rows]) ** 2).sum(-1) + 1e-4).sum()

    (g_full,) = torch.autograd.grad(loss_of(slice(0, 6)), W)
    lo, hi = shard_bounds(6, rank, world)
    (g_shard,) = torch.autograd.grad(loss_of(slice(lo, hi)), W)
    flat = torch.zeros(100003)  # several buckets, odd size
    flat[:35] = g_shard.flatten()
    flat[35:] = float(rank + 1)
    sync = GradSync(flat, bucket_bytes=64 * 1024)
    assert len(sync.buckets()) > 1 and sync.buckets()[-1][1] == flat.numel()
    sync.all_reduce()
    ok = torch.allclose(flat[:35].view(7, 5), g_full, rtol=1e-5, atol=1e-6) and bool((flat[35:] == 3.0).all())
    params = torch.full((10,), float(rank))
    broadcast_parameters(params, [])
    ok = ok and bool((params == 0).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gloo_two_ranks_sum_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover_without_overlap():
    for n in (1, 5, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    flat = torch.arange(10.0)
    GradSync(flat).all_reduce()
    assert torch.equal(flat, torch.arange(10.0))


def test_stage_slices_skip_the_frozen_trunk_body():
    """feature_extract and use_pretrained (util/model_utils.py:110-113 of the reference) freeze the ResNet body: its gradient elements are
    zero on every rank, so the staged exchange must cover the trainable segments only -- the replaced fc and the heads -- and every
    body stage must come out empty; with nothing frozen the stages tile the whole arena."""
    import torch.nn as nn

    class Trunk(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 4, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(4)
            self.layer1, self.layer2, self.layer3, self.layer4 = (nn.Sequential(nn.Conv2d(4, 4, 1, bias=False), nn.BatchNorm2d(4)) for _ in range(4))
            self.fc = nn.Linear(4, 6)

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.trunk = Trunk()
            self.head = nn.Linear(6, 7)

    m = Model()
    m._arena = ParamArena(m)
    full = stage_slices(m)
    spans = sorted(v for k, v in full.items() if k != "layer1")
    assert spans[0][0] == 0 and spans[-1][1] == m._arena.numel and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for name, p in m.trunk.named_parameters():
        if not name.startswith("fc."):
            p.requires_grad = False
    frozen = stage_slices(m)
    for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
        assert frozen[name][1] == frozen[name][0], name
    lo, hi = frozen["fc"]
    off = {id(p): o for p, o in zip(m._arena.params, m._arena.offsets)}
    assert lo == off[id(m.trunk.fc.weight)] and hi == m._arena.numel
    flat = torch.ones(m._arena.numel)
    sync = GradSync(flat, reduce_single=False)
    sync._slices, sync._seen = frozen, set()
    for name in ("fc", "layer4", "layer3", "layer2", "layer1", "stem"):
        sync.stage_done(name)
    sync.finish()
