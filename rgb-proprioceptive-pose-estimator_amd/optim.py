"""Adam over the flat parameter arena: one HIP kernel per trainable segment instead of ~170 per-tensor
update chains.  Same hyper-parameter defaults and update rule as torch.optim.Adam, which the reference
constructs at scripts/train_model.py:228 (no weight decay, no amsgrad); parameters whose gradient is
identically zero (e.g. the unused depth head) do not move, matching torch's `grad is None` skip.
"""
import torch

from . import ops
from ._lib import lib
from .params import arena_of


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """capturable: keep the step count (for the bias corrections) on the DEVICE, so that a captured hipGraph of the whole
        train step (util.learn_utils.GraphedTrainStep) advances it at every replay; a host-side count would be frozen at its
        capture-time value.  Same update rule either way."""
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        self._dev_state = None
        self._step = 0
        self._m = self._v = None
        self._arena = None

    def _ensure(self):
        params = [p for g in self.param_groups for p in g["params"]]
        arena = arena_of(params)
        if arena is None:
            raise RuntimeError("FusedAdam needs the model's flat parameter arena: run one forward on the device "
                               "(or call model._materialize) before the first step")
        if arena is not self._arena:
            self._arena = arena
            if self._dev_state is not None:
                self._dev_state = self._dev_state.to(arena.flat.device)
            amp_sd = getattr(self, "_amp_restore", None)
            if amp_sd is not None and getattr(arena, "loss_scaler", None) is not None:
                arena.loss_scaler.load_state_dict(amp_sd)
                self._amp_restore = None
            loaded = self._m is not None and self._m.numel() == arena.flat.numel()   # moments restored by load_state_dict
            self._m = self._m.to(arena.flat.device) if loaded else torch.zeros_like(arena.flat)
            self._v = self._v.to(arena.flat.device) if loaded else torch.zeros_like(arena.flat)
        return arena

    def zero_grad(self, set_to_none=True):
        """The HIP backward overwrites every gradient view, so there is nothing to clear on the device; this only tells the arena
        that the next backward starts fresh instead of accumulating (reference loop: util/learn_utils.py:152)."""
        arena = self._arena
        if arena is None:
            params = [p for g in self.param_groups for p in g["params"]]
            arena = arena_of(params)
        if arena is not None:
            arena.zero_grad()
        return None

    @torch.no_grad()
    def step(self, closure=None):
        arena = self._ensure()
        self._step += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        scaler = getattr(arena, "loss_scaler", None)
        if scaler is not None:
            # fp16 compute (amp.py): unscale + finite check, scale update and the skip decision all stay on the device; the
            # bias-correction step count is the device's count of steps actually taken (self._step counts calls)
            st = scaler.unscale_and_update(arena.grad)
            s = ops._stream()
            for lo, hi in arena.trainable_segments():
                lib.rpe_adam_step_amp(ops._p(arena.flat[lo:hi]), ops._p(arena.grad[lo:hi]), ops._p(self._m[lo:hi]), ops._p(self._v[lo:hi]), hi - lo,
                                      g["lr"], b1, b2, g["eps"], ops._p(st), s)
            return None
        if self.capturable:
            # device-side step count: the same state block the loss scaler uses (amp.py), with scale 1 and no unscale pass
            if self._dev_state is None or self._dev_state.device != arena.flat.device:
                st = torch.zeros(8, dtype=torch.float32)
                st[0] = st[1] = 1.0
                st[5] = float(self._step - 1)
                self._dev_state = st.to(arena.flat.device)
            s = ops._stream()
            lib.rpe_amp_update(ops._p(self._dev_state), 1.0, 1.0, 1 << 30, s)   # steps += 1 (found_inf is never set)
            for lo, hi in arena.trainable_segments():
                lib.rpe_adam_step_amp(ops._p(arena.flat[lo:hi]), ops._p(arena.grad[lo:hi]), ops._p(self._m[lo:hi]), ops._p(self._v[lo:hi]), hi - lo,
                                      g["lr"], b1, b2, g["eps"], ops._p(self._dev_state), s)
            return None
        for lo, hi in arena.trainable_segments():
            ops.adam_step(arena.flat[lo:hi], arena.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], g["lr"], b1, b2, g["eps"], self._step)
        return None

    def state_dict(self):
        """Flat first / second moments in arena order (= model.parameters() order, 16-byte padded segments) + the step count.
        The reference saves no optimizer state (util/learn_utils.py:211-241); this is what a resumable checkpoint adds."""
        sd = {"step": self._step, "m": self._m, "v": self._v, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        # device-side step count (capturable / fp16 steps) and the loss scaler's state, so that a resumed run continues both
        if self._dev_state is not None:
            sd["dev_state"] = self._dev_state.detach().cpu().clone()
        scaler = getattr(self._arena, "loss_scaler", None) if self._arena is not None else None
        if scaler is not None:
            sd["amp"] = scaler.state_dict()
        return sd

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        self._m = None if sd["m"] is None else sd["m"].clone()
        self._v = None if sd["v"] is None else sd["v"].clone()
        self._arena = None   # re-attached (and the moments moved to its device) at the next step
        self._dev_state = sd["dev_state"].clone() if sd.get("dev_state") is not None else None
        # fp16: the loss scaler must hold the checkpointed scale BEFORE the first backward after the resume multiplies the output
        # gradients by it (round 2 handed it over inside the first step(), i.e. after that backward had used a fresh 2^12: the first
        # step's gradients were off by the ratio of the two scales and went into Adam's moments).  The scaler object belongs to the
        # model: load it now if the arena exists, else leave it on the parameters for the model's _materialize to install.
        amp_sd = sd.get("amp")
        self._amp_restore = None
        if amp_sd is not None:
            params = [p for g in self.param_groups for p in g["params"]]
            arena = arena_of(params)
            scaler = getattr(arena, "loss_scaler", None) if arena is not None else None
            if scaler is not None:
                scaler.load_state_dict(amp_sd)
            else:
                self._amp_restore = amp_sd
                for p in params:
                    p._rpe_pending_amp = amp_sd
        for g, sg in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in sg.items() if k != "params"})
