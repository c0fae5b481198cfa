"""Dynamic loss scaling for the fp16 compute path (BASELINE config C5: "TDO_v2 ... fp16 + MFMA conv").

The reference trains in fp32 (scripts/train_model.py:228, util/learn_utils.py:152-179) and has no counterpart; fp16
activations / activation gradients have 5 exponent bits, so the backward signal is multiplied by a power of two at the
model output and every parameter gradient divided by it again before Adam.  The whole protocol runs on the device:

    backward:   d(loss)/d(outputs) *= scale                       (models/_core.py, _ModelFn.backward)
    [SUM all-reduce of the still-scaled gradients]                (dist.GradSync: linear, so scaling commutes)
    optimizer:  grads *= 1/scale, found_inf = any non-finite      (rpe_amp_unscale)
                found_inf ? scale *= backoff, skip : steps += 1   (rpe_amp_update; growth every `growth_interval` finite steps)
                Adam, skipped on the device when `skip` is set    (rpe_adam_step_amp)

so a step never synchronises the host, and replicas take the same decision because they test the REDUCED gradients.
bf16 and fp32 compute need none of this (8 exponent bits) and never construct a LossScaler.
"""
import torch

from . import ops
from ._lib import lib


class LossScaler:
    SCALE, INV, FOUND_INF, SKIP, STREAK, STEPS = range(6)

    def __init__(self, init_scale=2.0 ** 12, growth_factor=2.0, backoff_factor=0.5, growth_interval=200):
        self.init_scale, self.growth, self.backoff, self.interval = float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = None      # 8 device floats, created on first use

    def _ensure(self, device):
        if self.state is None:
            s = torch.zeros(8, dtype=torch.float32)
            s[self.SCALE], s[self.INV] = self.init_scale, 1.0 / self.init_scale
            self.state = s.to(device)
        elif self.state.device != device:
            self.state = self.state.to(device)   # e.g. restored from a checkpoint on the host
        return self.state

    def scale_tensor(self, device):
        """0-d device tensor holding the current scale (multiply the output gradients by it)."""
        return self._ensure(device)[self.SCALE]

    def unscale_and_update(self, flat_grad):
        """grads *= 1/scale with the finite check, then the scale / skip / step-count update.  Returns the state tensor."""
        st = self._ensure(flat_grad.device)
        s = ops._stream()
        lib.rpe_amp_unscale(ops._p(flat_grad), flat_grad.numel(), ops._p(st), s)
        lib.rpe_amp_update(ops._p(st), self.growth, self.backoff, self.interval, s)
        return st

    # -- host-side views (each synchronises; for logging / tests / checkpoints only) ----------------
    def get_scale(self):
        return self.init_scale if self.state is None else float(self.state[self.SCALE].item())

    def steps_taken(self):
        return 0 if self.state is None else int(self.state[self.STEPS].item())

    def state_dict(self):
        return {"state": None if self.state is None else self.state.detach().cpu().clone(), "init_scale": self.init_scale, "growth": self.growth,
                "backoff": self.backoff, "interval": self.interval}

    def load_state_dict(self, sd):
        self.init_scale, self.growth, self.backoff, self.interval = sd["init_scale"], sd["growth"], sd["backoff"], sd["interval"]
        self.state = None if sd["state"] is None else sd["state"].clone()   # moved to the device by _ensure at the next use
