"""Batch layout and device staging for the train step.

The reference's MultiEpisodeDataset (util/data_utils.py:10-204) steps a Robosuite/MuJoCo simulator to
produce episodes; that simulator and its CPU image preprocessing are out of scope (SURVEY.md section 2,
row 7).  What the train step depends on is kept: the time-major layout (`__getitem__(t)` returns all
episodes at timestep t, util/data_utils.py:62-73), the 6-tuple, `refresh_data`, `env.horizon`, and
`standardize_quat`.  SyntheticEpisodeDataset fills the same tensors with seeded Robosuite-shaped data
(ImageNet-normalised uint8 noise images, workspace-bounded positions, unit quaternions with w >= 0,
proprioception = truth + N(0, noise_scale I) with the quaternion renormalised, util/data_utils.py:162-176),
generated directly in HBM.
"""
import types

import torch
from torch.utils.data import Dataset

MOTIONS = {"random", "up", "up_random"}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size, out_size):
    """Tap tables of Pillow's antialiased 8-bit bilinear resample along one axis (what `Resize(256)` on a PIL image runs,
    util/data_utils.py:48-54 of the reference): (bounds [out, 2] int32 = first input index and tap count, weights [out, ksize]
    int32 in 22-bit fixed point).  Host side, double precision, exactly Pillow's arithmetic (Resample.c precompute_coeffs +
    normalize_coeffs_8bpc); the device kernels only multiply-accumulate with them (csrc/norm.hip resize_*_kernel)."""
    import math
    import numpy as np
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)       # (int) truncation of non-negative values
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    t = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) / filterscale)
    w = np.where((t < 1.0) & (taps < xmax[:, None]), 1.0 - t, 0.0)
    ww = w.sum(1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    kk = np.floor(0.5 + w * (1 << PIL_PRECISION_BITS)).astype(np.int32)   # weights are >= 0 for the triangle filter: (int)(0.5 + v)
    kk = np.where(taps < xmax[:, None], kk, 0).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


def resized_hw(h, w, size=256):
    """torchvision.transforms.Resize(int) geometry: the shorter side becomes `size`, the longer int(size * long / short)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def crop_origin(h, w, ch, cw):
    """torchvision.transforms.CenterCrop: int(round((h - ch) / 2.0)), Python rounding (half to even)."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def standardize_quat(quat):
    """(x,y,z,w) quaternion with a non-negative w (reference: util/data_utils.py:207-211)."""
    return -quat if quat[-1] < 0 else quat


def random_poses(lead, generator, device):
    pos = torch.rand(*lead, 3, generator=generator, device=device)
    pos = pos * torch.tensor([0.7, 0.7, 0.5], device=device) + torch.tensor([-0.35, -0.35, 0.8], device=device)
    q = torch.randn(*lead, 4, generator=generator, device=device)
    q = q / q.norm(dim=-1, keepdim=True)
    q = torch.where(q[..., 3:4] < 0, -q, q)
    return torch.cat([pos, q], dim=-1)


def synthetic_batch(lead, seed, hw=224, with_depth=False, noise_scale=0.001, device="cuda"):
    """Seeded Robosuite-shaped batch with leading dims `lead` ((N,) or (S, N)), created on `device`."""
    lead = tuple(lead)
    g = torch.Generator(device=device).manual_seed(int(seed))
    u8 = torch.randint(0, 256, (*lead, 3, hw, hw), generator=g, device=device, dtype=torch.uint8)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(3, 1, 1)
    img = (u8.float() / 255.0 - mean) / std
    depth = torch.rand(*lead, 1, hw, hw, generator=g, device=device) if with_depth else None
    x0, x1, obj = (random_poses(lead, g, device) for _ in range(3))
    x0bar = x0 + (noise_scale ** 0.5) * torch.randn(*lead, 7, generator=g, device=device)
    qb = x0bar[..., 3:]
    x0bar = torch.cat([x0bar[..., :3], qb / qb.norm(dim=-1, keepdim=True)], dim=-1)
    return {"img": img, "depth": depth, "x0bar": x0bar, "x0": x0, "x1": x1, "obj": obj}


class SyntheticEpisodeDataset(Dataset):
    """MultiEpisodeDataset-shaped source of seeded synthetic episodes, resident on `device`."""

    def __init__(self, horizon=20, use_depth=False, obj_name=None, is_two_arm=False, motion="random", seed=1234, hw=224,
                 device="cuda", env_name="Synthetic"):
        if motion not in MOTIONS:
            raise ValueError("Invalid motion specified. {} supported, {} requested.".format(MOTIONS, motion))
        self.data = None
        self.obj_name = obj_name
        self.use_depth = use_depth
        self.is_two_arm = is_two_arm
        self.motion = motion
        self.seed, self.hw, self.device = seed, hw, device
        self._refreshes = 0
        # train() only reads type(env).__name__ and env.horizon (util/learn_utils.py:84-89)
        self.env = type(env_name, (), {})()
        self.env.horizon = horizon

    def __len__(self):
        return self.data["measurement_self"].size(1)

    def __getitem__(self, index):
        d = self.data
        img = d["imgs"][:, index]
        depth = d["depths"][:, index] if self.use_depth else torch.empty(0, device=img.device)
        x0bar = d["measurement_self"][:, index]
        x0 = d["true_self"][:, index]
        x1 = d["true_other"][:, index] if self.is_two_arm else torch.empty_like(x0)
        obj = d["true_obj"][:, index] if self.obj_name is not None else torch.empty_like(x0)
        return img, depth, x0bar, x0, x1, obj

    def refresh_data(self, num_episodes, camera_name=None, noise_scale=0.001):
        b = synthetic_batch((num_episodes, self.env.horizon), self.seed + self._refreshes, self.hw, self.use_depth, noise_scale, self.device)
        self._refreshes += 1
        self.data = {"imgs": b["img"], "depths": b["depth"], "measurement_self": b["x0bar"], "true_self": b["x0"],
                     "true_other": b["x1"], "true_obj": b["obj"]}

    def chunk(self, t0, length):
        """Time-major chunk (S, N, ...) of timesteps [t0, t0+length): what DataLoader(batch_size=S, shuffle=False) stacks."""
        d = self.data
        sl = slice(t0, t0 + length)
        tm = lambda x: x[:, sl].transpose(0, 1).contiguous()
        img = tm(d["imgs"])
        depth = tm(d["depths"]) if self.use_depth else None
        x1 = tm(d["true_other"]) if self.is_two_arm else None
        obj = tm(d["true_obj"]) if self.obj_name is not None else None
        return img, depth, tm(d["measurement_self"]), tm(d["true_self"]), x1, obj


class FramePrefetcher:
    """Host -> device staging of raw simulator frames, double buffered.

    replaces: the per-tensor synchronous pageable `.cuda()` of util/learn_utils.py:130-138 fed by the CPU transform
    (util/data_utils.py:48-54).  `batches` yields tuples of CPU tensors; uint8 frame tensors stay uint8 -- 50 MB instead of
    154 MB of fp32 per 256 frames -- and are cropped / normalised on the device by the trunk (`rpe_stage_frames_u8`).  Every
    tensor goes through a pinned staging buffer and an asynchronous copy on a side stream into one of `depth` device slots;
    the consumer's stream waits for that copy only, so batch k+1 crosses PCIe while batch k trains.

        for frames, x0bar, target in FramePrefetcher(loader, device):
            loss = criterion(model(frames, None, x0bar), target)
    """

    def __init__(self, batches, device="cuda", depth=2):
        self.batches, self.device, self.depth = batches, torch.device(device), max(2, int(depth))
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = [None] * self.depth   # per slot: list of pinned host buffers
        self._dev = [None] * self.depth      # per slot: list of device buffers
        self._ready = [None] * self.depth    # per slot: copy-finished event
        self._free = [None] * self.depth     # per slot: consumer-finished event (the slot may be overwritten after it)

    def _stage(self, slot, batch):
        items = list(batch) if isinstance(batch, (tuple, list)) else [batch]
        if self._dev[slot] is None or len(self._dev[slot]) != len(items) or any(
                (d is None) != (t is None) or (t is not None and (d.shape != t.shape or d.dtype != t.dtype or
                                                                   ((self._pinned[slot][i] is None) != t.is_pinned())))
                for i, (d, t) in enumerate(zip(self._dev[slot], items))):
            self._pinned[slot] = [None if (t is None or t.is_pinned()) else torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in items]
            self._dev[slot] = [None if t is None else torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in items]
        if self._free[slot] is not None:
            self.stream.wait_event(self._free[slot])      # the previous consumer of this slot is done with it
        # The slot's pinned staging buffers are the SOURCE of its previous asynchronous H2D copy, which may not even have
        # started yet (it queues behind `_free[slot]` on the copy stream while the training thread runs ahead of the GPU):
        # the host must not overwrite them before that copy has finished.
        if self._ready[slot] is not None and any(p is not None for p in self._pinned[slot]):
            self._ready[slot].synchronize()
        src = []
        for p, t in zip(self._pinned[slot], items):
            if t is None or t.is_pinned():
                src.append(t)                               # already page-locked (DataLoader(pin_memory=True)): DMA straight from it
            else:
                p.copy_(t)                                  # host memcpy into the pinned buffer (on the caller's thread: keep
                src.append(p)                               # loaders pinning, or this copy is what the GPU waits for)
        self._src_keep = getattr(self, "_src_keep", [None] * self.depth)
        self._src_keep[slot] = src                          # sources stay alive until the slot is staged again
        with torch.cuda.stream(self.stream):
            for d, p in zip(self._dev[slot], src):
                if p is not None:
                    d.copy_(p, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._ready[slot] = ev

    def __iter__(self):
        it = iter(self.batches)
        slot, pending = 0, []
        try:
            for _ in range(self.depth - 1):
                self._stage(slot, next(it))
                pending.append(slot)
                slot = (slot + 1) % self.depth
        except StopIteration:
            pass
        while pending:
            cur = pending.pop(0)
            try:
                self._stage(slot, next(it))
                pending.append(slot)
                slot = (slot + 1) % self.depth
            except StopIteration:
                pass
            torch.cuda.current_stream(self.device).wait_event(self._ready[cur])
            yield tuple(self._dev[cur])
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self._free[cur] = done
