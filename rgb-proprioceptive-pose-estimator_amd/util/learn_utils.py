"""Training loop of the pose estimators on the MI355X path.

Same signature, phases, criterion dict, statistics and checkpoint naming as train() of the reference
(util/learn_utils.py:21-255); what changes is underneath:
  * batches are time-major slices of data already resident in HBM when the dataset offers `chunk()`; a dataset with only
    the reference's `__getitem__` contract is stacked the way its DataLoader would and staged through pinned, double-buffered
    asynchronous copies (no per-tensor pageable .cuda(), util/learn_utils.py:75-76,130-138);
  * the per-step "val" metric is reduced on the device and only read back once per phase (the reference
    synchronises twice per step: models/losses.py:99-113 and util/learn_utils.py:182);
  * with torch.distributed initialised, episodes are sharded over ranks and the flat gradient buffer is
    SUM-all-reduced (RCCL) between backward and the optimizer step.
rollout() of the reference needs the simulator and is out of scope.
"""
import copy
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from ..dist import GradSync, broadcast_parameters, shard_bounds


def _writer(logging):
    if not logging:
        return None
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter()
    except Exception:  # tensorboard is optional here
        return None


def _host_chunks(dataset, horizon, seq, use_depth):
    """What `DataLoader(dataset, batch_size=S, shuffle=False)` yields for a reference-shaped dataset (util/learn_utils.py:75-76,
    util/data_utils.py:62-73): `dataset[t]` is the 6-tuple of all episodes at timestep t; S consecutive timesteps are stacked
    time-major (S, N, ...).  Fields the model never reads (the reference fills them with torch.empty garbage) are dropped here
    instead of being copied to the device."""
    for t0 in range(0, horizon, seq):
        items = [dataset[t] for t in range(t0, min(t0 + seq, horizon))]
        cols = list(zip(*items))
        stack = lambda c: torch.stack([torch.as_tensor(x) for x in c], 0)
        img, depth, x0bar, x0, x1, obj = cols
        yield (stack(img), stack(depth) if use_depth else None, stack(x0bar), stack(x0), stack(x1), stack(obj))


def _chunks(dataset, horizon, seq, use_depth):
    """Time-major device batches of one phase.  Datasets that keep their episodes in HBM offer `chunk(t0, S)` (the synthetic one
    does); any other dataset with the reference's contract -- `__len__`, `__getitem__(t)` -> 6-tuple of host tensors -- goes
    through pinned, double-buffered asynchronous staging (FramePrefetcher), replacing the reference's per-tensor synchronous
    pageable `.cuda()` (util/learn_utils.py:130-138)."""
    if hasattr(dataset, "chunk"):
        for t0 in range(0, horizon, seq):
            yield dataset.chunk(t0, min(seq, horizon - t0))
        return
    from .data_utils import FramePrefetcher
    first = dataset[0][0]
    if torch.as_tensor(first).is_cuda:   # device-resident tensors behind a plain __getitem__: just stack
        yield from _host_chunks(dataset, horizon, seq, use_depth)
        return
    yield from FramePrefetcher(_host_chunks(dataset, horizon, seq, use_depth), torch.device("cuda", torch.cuda.current_device()))


def train_step(model, batch, criterion, optimizer, train_obj_pose, phase="train", grad_sync=None):
    """One iteration of the reference's hot loop (util/learn_utils.py:152-184).  Returns device scalars
    (loss, pos_err, ori_err) -- nothing is synchronised to the host."""
    img, depth, x0bar, x0, x1, obj = batch
    optimizer.zero_grad()
    with torch.set_grad_enabled(phase == "train"):
        if train_obj_pose:
            obj_out = model(img, depth, x0bar)
            loss = criterion["obj_loss"](obj_out, obj)
            pos_err, ori_err = criterion["val_loss"].forward_device(obj_out, obj)
        else:
            x0_out, x1_out = model(img, depth, x0bar)
            loss = criterion["x0_loss"](x0_out, x0) + criterion["x1_loss"](x1_out, x1)
            pos_err, ori_err = criterion["val_loss"].forward_device(x1_out, x1)
        if phase == "train":
            loss.backward()
            if grad_sync is not None:
                if grad_sync.staged:   # slices were launched under the backward (dist.GradSync.attach): just wait
                    grad_sync.finish()
                else:
                    grad_sync.all_reduce()
            optimizer.step()
    return loss.detach(), pos_err, ori_err


def _graph_keepalive(model):
    """What a captured graph has baked addresses of, beyond its own tensors: the trunk plans (native engine + workspace) that exist
    at capture time and the per-device weight-gradient scratch.  A later eager call may evict a plan from the trunk's LRU table or
    replace the scratch by a larger buffer; holding these references keeps the captured addresses alive for as long as the graph
    object lives (the eager path simply continues on its new buffers)."""
    from .. import ops
    return (list(model.trunk._plans.values()), list(ops._SCRATCH.values()))


class GraphedTrainStep:
    """One train step (forward -> loss -> on-device val metrics -> backward -> Adam) captured ONCE into a hipGraph and replayed.

    The step is ~330 kernel launches issued from Python + C++ through ctypes; replaying one graph removes the launch gaps on the
    device and the host work between them (HIP streams and graphs instead of a tracing compiler).  Everything a replay touches
    lives at a fixed address: the inputs are copied into static tensors, the trunk plan's workspace and the weight-gradient
    scratch are allocated before the capture, temporaries come from the graph's private pool, and the optimizer's step count is
    kept on the device (FusedAdam(capturable=True)).  Single-process only: with a process group the step stays eager (RCCL
    collectives inside a captured graph are not exercised here).

        step = GraphedTrainStep(model, criterion, optimizer, train_obj_pose=True, example_batch=batch)
        loss, pos_err, ori_err = step(batch)          # device scalars, valid until the next call
    """

    def __init__(self, model, criterion, optimizer, train_obj_pose, example_batch, warmup=3):
        if dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep is single-process; data-parallel steps run eagerly")
        if not getattr(optimizer, "capturable", False):
            raise RuntimeError("GraphedTrainStep needs FusedAdam(..., capturable=True): the step count must live on the device")
        self.model, self.criterion, self.optimizer, self.train_obj_pose = model, criterion, optimizer, train_obj_pose
        self.static = tuple(None if t is None else t.clone() for t in example_batch)
        model.train()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up on a side stream: plans, workspaces and scratch buffers exist before the capture
            for _ in range(warmup):
                train_step(model, self.static, criterion, optimizer, train_obj_pose, "train", None)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = train_step(model, self.static, criterion, optimizer, train_obj_pose, "train", None)
        self.warmup_steps = warmup   # optimizer steps taken while building (the capture itself does not execute anything)
        self._keep = _graph_keepalive(model)

    def __call__(self, batch):
        for dst, src in zip(self.static, batch):
            if dst is not None and dst is not src:   # (fill `self.static` in place to skip the copy)
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # The replay re-ran the captured weight-packing launch, the optimizer and the BN running-statistics updates behind the
        # host's back: every plan's cached weight copies (the BN-folded inference copies above all) are stale now, exactly as
        # after an eager step.  Bumping the trunk's weight version makes the next eval forward -- on this plan or any other --
        # re-fold before it runs.
        self.model.trunk.weights_changed()
        return self.out


class GraphedRolloutFrame:
    """One eval-mode rollout frame (BN folded into the convs, LSTM state carried in place on the device) as a hipGraph:
    the per-frame path of rollout() (util/learn_utils.py:322-323,342,446 of the reference) is launch-bound at batch 1."""

    def __init__(self, model, img, depth, x0bar, warmup=2, calibrate=8):
        self.model = model
        model.eval()
        self.img, self.x0bar = img.clone(), x0bar.clone()
        self.depth = None if depth is None else depth.clone()
        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calibrate):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / calibrate * 1e3

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.replay_ms = self.eager_ms = None
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model(self.img, self.depth, self.x0bar)
            # A replay is worth it only where the runtime launches the graph cheaply: both forms are timed over `calibrate` frames and the
            # slower one is dropped (see below).  The eager frames come BEFORE the capture: the captured frame bakes in the addresses of
            # the carried LSTM state as the last eager frame left them.  (Like the warm-up frames they advance that state: the caller
            # starts its episode with reset_initial_state afterwards.)
            if calibrate > 0:
                self.eager_ms = timed(lambda: model(self.img, self.depth, self.x0bar))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(self.img, self.depth, self.x0bar)
        self._keep = _graph_keepalive(model)
        # On the round-3 boxes a captured frame that forks to the engine's second stream replayed in 0.56 ms in one process and in
        # 3.2 ms in the next (hipGraphLaunch itself 2.6 ms of host time; round 2's tree behaves the same there:
        # profiles/r03_rollout_latency.txt); the engine now keeps a captured inference frame on ONE stream (0.45 ms, reliably), and
        # this check stays as the guard: replay slower than the eager frame -> the eager frame serves.
        if calibrate > 0:
            self.graph.replay()
            self.replay_ms = timed(self.graph.replay)
            if self.replay_ms > self.eager_ms:
                self.graph = None
                self._keep = None

    @property
    def replaying(self):
        return self.graph is not None

    def __call__(self, img, depth, x0bar):
        if self.graph is None:   # (the eager frame measured faster than the replay on this box)
            dev = self.img.device
            with torch.no_grad():
                return self.model(img.to(dev, non_blocking=True), None if depth is None else depth.to(dev, non_blocking=True), x0bar.to(dev, non_blocking=True))
        self.img.copy_(img, non_blocking=True)
        self.x0bar.copy_(x0bar, non_blocking=True)
        if self.depth is not None:
            self.depth.copy_(depth, non_blocking=True)
        self.graph.replay()
        return self.out


def train(model, dataset, criterion, optimizer, num_epochs, num_train_episodes_per_epoch, num_val_episodes_per_epoch, params, device,
          save_path='default', save_model=True, logging=True, *, save_optimizer=False):
    """See the module docstring.  Returns (model with the best validation weights, best validation loss).
    save_optimizer (addition; the reference saves weights only): also write `<save_path>.optim` with the optimizer state of the
    best-validation epoch so that a run can be resumed (`optimizer.load_state_dict(torch.load(path))`)."""
    train_obj_pose = hasattr(model, "object_name")
    dt_string = datetime.now().strftime("%d-%m-%Y_%H-%M-%S")
    since = time.time()
    best_model = copy.deepcopy(model.state_dict())
    best_err = np.inf
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    writer = _writer(logging and rank == 0)
    if device == "cpu":
        raise RuntimeError("train(): the pose train step runs on the MI355X HIP path only (device='cuda:N'); there is no CPU fallback")
    model.cuda()
    # Build the flat parameter arena now, so replicas can be made identical and the gradient reduction attached BEFORE the
    # first optimizer step (Adam moments would otherwise diverge between ranks).
    model._materialize(torch.device("cuda", torch.cuda.current_device()))
    grad_sync = None
    if world > 1:
        broadcast_parameters(model._arena.flat, list(model.buffers()))
        grad_sync = GradSync(model._arena.grad).attach(model)
    seq = model.sequence_length if model.requires_sequence else 1
    fname = "{}_{}_{}hzn_{}ep_{}.pth".format(type(model).__name__, type(dataset.env).__name__, dataset.env.horizon,
                                             num_epochs * num_train_episodes_per_epoch, dt_string)
    if save_model and rank == 0:
        print("\nFile name saved:\n{}\n".format(fname))
    for epoch in range(num_epochs):
        if logging and rank == 0:
            print("\n" + "-" * 10 + "\nEpoch {}/{}\n".format(epoch, num_epochs - 1) + "-" * 10)
        for phase in ["train", "val"]:
            num_episodes = num_train_episodes_per_epoch if phase == "train" else num_val_episodes_per_epoch
            model.train() if phase == "train" else model.eval()
            lo, hi = shard_bounds(num_episodes, rank, world)
            dataset.refresh_data(hi - lo, params["camera_name"], params["noise_scale"])
            model.reset_initial_state(hi - lo)
            sums = torch.zeros(3, dtype=torch.float64, device="cuda")
            horizon = len(dataset)
            for img, depth, x0bar, x0, x1, obj in _chunks(dataset, horizon, seq, model.use_depth if hasattr(model, "use_depth") else False):
                if not model.requires_sequence:  # the reference squeezes the leading batch-of-1 dim (learn_utils.py:141-149)
                    img, x0bar, x0 = img[0], x0bar[0], x0[0]
                    depth = None if depth is None else depth[0]
                    x1 = None if x1 is None else x1[0]
                    obj = None if obj is None else obj[0]
                loss, pe, oe = train_step(model, (img, depth, x0bar, x0, x1, obj), criterion, optimizer, train_obj_pose, phase, grad_sync)
                sums += torch.stack([loss.double(), pe.double(), oe.double()])
            if world > 1:
                dist.all_reduce(sums)
            tot = sums.tolist()  # the one host synchronisation of the phase
            denom = horizon * num_episodes
            epoch_loss, epoch_pos_err, epoch_ori_err = tot[0] / denom, tot[1] / denom, tot[2] / denom
            time_elapsed = time.time() - since
            if writer is not None:
                tag = "train" if phase == "train" else "val"
                writer.add_scalar("Loss/" + tag, epoch_loss, epoch)
                writer.add_scalar("Err_pos/" + tag, epoch_pos_err, epoch)
                writer.add_scalar("Err_ori/" + tag, epoch_ori_err, epoch)
            if logging and rank == 0:
                print('{} Loss: {:.4f}, PosErr: {:.4f}, OriErr: {:.4f}. Time elapsed = {:.0f}m {:.0f}s'.format(
                    phase, epoch_loss, epoch_pos_err, epoch_ori_err, time_elapsed // 60, time_elapsed % 60))
            if phase == "val" and epoch_loss < best_err:
                best_err = epoch_loss
                best_model = copy.deepcopy(model.state_dict())
                if save_model and rank == 0:
                    if save_path == 'default':
                        save_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "log", "runs", fname)
                    os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
                    torch.save(model.state_dict(), save_path)
                    if save_optimizer:
                        torch.save(optimizer.state_dict(), save_path + ".optim")
    if logging and rank == 0:
        time_elapsed = time.time() - since
        print('-' * 10)
        print('Training completed in {:.0f}m {:.0f}s'.format(time_elapsed // 60, time_elapsed % 60))
        print('Best val Err: {:.4f}'.format(best_err))
    model.load_state_dict(best_model)
    return model, best_err
