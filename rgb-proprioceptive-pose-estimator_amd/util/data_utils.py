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
