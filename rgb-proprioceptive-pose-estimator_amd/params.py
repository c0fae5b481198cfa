"""Flat parameter / gradient arena.

All parameters of a model become views into ONE fp32 buffer, and every parameter gets a
gradient view into a second buffer of the same layout.  The HIP kernels write gradients
straight into those views (no autograd accumulation pass), Adam is one kernel over the flat
buffers, and the data-parallel all-reduce is a few large RCCL calls over contiguous slices.
"""
import torch


def _pad4(n):
    return (n + 3) // 4 * 4


class ParamArena:
    def __init__(self, module):
        seen, params = set(), []
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("ParamArena needs all parameters on one device in fp32")
        self.params = params
        self.offsets = []
        total = 0
        for p in params:
            self.offsets.append(total)
            total += _pad4(p.numel())  # 16-byte aligned segments
        self.numel = total
        self._dirty = False       # a backward has written `grad` and nothing has zeroed it since (see accumulating())
        self.loss_scaler = None   # set by fp16 models: gradients in `grad` carry its scale until the optimizer unscales them
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(params, self.offsets):
            n = p.numel()
            dst = self.flat[off:off + n].as_strided(p.shape, p.stride())
            dst.copy_(p.data)
            p.data = dst
            p._rpe_grad = self.grad[off:off + n].as_strided(p.shape, p.stride())
            p._rpe_arena = (self, off, n)

    def is_current(self):
        """False once something (e.g. module.to()) re-allocated a parameter outside the arena."""
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + self.numel * 4
        return all(base <= p.data_ptr() < end for p in self.params)

    def accumulating(self):
        """True when the next backward must ADD to the gradients instead of replacing them: torch semantics are that .grad
        accumulates across backward() calls until the optimizer's zero_grad().  The kernels overwrite their gradient views, so
        the model's backward carries the old values over only in this case; the reference loop zeroes every iteration
        (util/learn_utils.py:152) and never pays for it.  `zero_grad(set_to_none=True)` of a torch optimizer is recognised by
        the .grad attributes it cleared."""
        if not self._dirty:
            return False
        return all(p.grad is not None for p in self.params if p.requires_grad)

    def zero_grad(self):
        self._dirty = False

    def publish_grads(self):
        """Expose the gradient views as .grad (what torch optimizers and checkpoints read)."""
        for p in self.params:
            if p.requires_grad:
                if p.grad is not p._rpe_grad:
                    p.grad = p._rpe_grad
            else:
                p.grad = None

    def trainable_segments(self):
        """Merged [start, end) ranges of the flat buffer that belong to requires_grad parameters."""
        segs = []
        for p, off in zip(self.params, self.offsets):
            if not p.requires_grad:
                continue
            end = off + _pad4(p.numel())
            if segs and segs[-1][1] == off:
                segs[-1][1] = end
            else:
                segs.append([off, end])
        return [tuple(s) for s in segs]


def arena_of(params):
    """The arena shared by all `params` (or None)."""
    arena = None
    for p in params:
        a = getattr(p, "_rpe_arena", None)
        if a is None:
            return None
        if arena is None:
            arena = a[0]
        elif arena is not a[0]:
            return None
    return arena
