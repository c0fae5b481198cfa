"""Shared machinery of the five pose estimators: device materialisation (flat parameter arena +
native trunk plan), the early-feature aux/depth heads, and the bridge into torch autograd.

The reference repeats the ResNet-import / bn1-hook / aux-head construction in four classes
(models/naive.py:188-253, models/time_sensitive.py:66-118,346-407,606-666); here it lives once.
Forward and backward of a model are explicit compositions of C-ABI ops (headops.py, engine.py):
autograd sees one node per model call, and parameter gradients are written by the kernels
directly into the arena's gradient views.
"""
import math
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..amp import LossScaler
from ..engine import ResNet50Trunk
from ..headops import AuxHeadOp, new_rows
from ..params import ParamArena
from ..util.model_utils import import_resnet

_DEFAULT_COMPUTE_DTYPE = torch.bfloat16


def set_default_compute_dtype(dtype):
    """torch.bfloat16 (fast path; fp32 accumulate + fp32 master weights), torch.float16 (same, with dynamic loss
    scaling: amp.py) or torch.float32 (parity path)."""
    global _DEFAULT_COMPUTE_DTYPE
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be torch.float32, torch.bfloat16 or torch.float16")
    _DEFAULT_COMPUTE_DTYPE = dtype


def default_compute_dtype():
    return _DEFAULT_COMPUTE_DTYPE


class Replicated(nn.Module):
    """Stands where the reference wraps a sub-module in nn.DataParallel (e.g. models/naive.py:224,253,274):
    contributes the same ``module.`` segment to state_dict keys and the same ``.module`` attribute.
    Replication itself is process-per-GPU + RCCL all-reduce (dist.py), not thread-per-replica."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class _AuxStack(nn.Sequential):
    """Conv2d(C->1, 1x1) -> MaxPool2d(2) -> Flatten : parameter holder for the aux head."""


class _DepthStack(nn.Sequential):
    """AvgPool2d(2) x k -> InstanceNorm2d(1, affine) -> Flatten : parameter holder for the depth head."""


def _make_aux(c):
    return _AuxStack(nn.Conv2d(c, 1, 1), nn.MaxPool2d(2), nn.Flatten())


def _make_depth(h, w):
    n_pool = int(math.log(224 ** 2 / (h * w // 4), 4))
    return _DepthStack(*([nn.AvgPool2d(2) for _ in range(n_pool)] + [nn.InstanceNorm2d(1, affine=True), nn.Flatten()]))


class _ModelFn(torch.autograd.Function):
    """One autograd node for a whole model call.  `anchor` is a trainable parameter: it only makes the
    node part of the graph -- gradients are written into the arena by the kernels, not returned."""

    @staticmethod
    def forward(ctx, model, anchor, img, depth, x0bar):
        ctx.model = model
        outs = model._forward_impl(img, depth, x0bar, save=True)
        ctx.n_out = len(outs)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *d_outs):
        model = ctx.model
        scaler = model.loss_scaler
        if scaler is not None:   # fp16 compute: the whole backward signal carries the loss scale (amp.py)
            sc = scaler.scale_tensor(next(d for d in d_outs if d is not None).device)
            d_outs = [None if d is None else d * sc for d in d_outs]
        arena = model._arena
        carry = arena.grad.clone() if arena.accumulating() else None   # a second backward() without zero_grad(): torch adds
        sync = getattr(model, "_grad_sync", None)
        model._backward_impl(list(d_outs))
        if carry is not None:
            if sync is not None:
                # staged data-parallel reduction: the slices of THIS backward are still being summed across replicas on the communication
                # stream; the carried gradients (already reduced by their own backward) are added once that is done -- GradSync.finish()
                sync.defer_add(carry)
            else:
                arena.grad.add_(carry)
        arena._dirty = True
        arena.publish_grads()
        return None, None, None, None, None


class PoseModelBase(nn.Module):
    """Common state: ``feature_net`` (ResNet-50 trunk), optional ``aux_nets`` / ``depth_nets``."""

    EARLY_SHAPE = (64, 112, 112)  # bn1 output for a 224x224 input; what the reference's dummy forward measures
    # hooked layer -> (C, H, W) of its output for the 224x224 dummy input the reference measures with (models/naive.py:213-216)
    HOOK_SHAPES = {0: (64, 112, 112), 9: (64, 112, 112), 1: (256, 56, 56), 2: (512, 28, 28), 3: (1024, 14, 14)}
    HOOK_ORDER = (0, 9, 1, 2, 3)

    def _init_features(self, num_resnet_layers, latent_dim, feature_extract, use_pretrained, feature_layer_nums, use_depth,
                       wrap, register_heads, compute_dtype):
        self.compute_dtype = compute_dtype or default_compute_dtype()
        self.loss_scaler = LossScaler() if self.compute_dtype == torch.float16 else None
        self.latent_dim = latent_dim
        self.use_depth = use_depth
        self.aux_latent_dim = 0
        self.early_features = None
        self.aux_nets = None
        self.depth_nets = None
        trunk, _ = import_resnet(num_resnet_layers, latent_dim, feature_extract, use_pretrained=use_pretrained,
                                 compute_dtype=self.compute_dtype)
        self.feature_net = trunk  # registered first, as in the reference, so state_dict order matches
        if trunk.expansion != 4:   # BasicBlock trunk (resnet18): the layer outputs have planes, not 4 x planes, channels
            self.HOOK_SHAPES = {k: ((c // 4 * trunk.expansion) if k in (1, 2, 3) else c, h, w) for k, (c, h, w) in self.HOOK_SHAPES.items()}
        self._hooks = []
        if feature_layer_nums is not None:
            layers = list(feature_layer_nums)
            for layer in layers:
                if layer == 4:
                    # the reference sizes its fc input with H*W//4 = 12 columns for layer4's 7x7 map while MaxPool2d(2) yields
                    # 3x3 = 9 (models/naive.py:243): its own forward raises a shape error for this hook
                    raise ValueError("feature_layer_nums: a hook on layer4 cannot run in the reference either (aux dim 7*7//4 = 12 vs 9 pooled features)")
                if layer not in self.HOOK_SHAPES:
                    raise ValueError("feature_layer_nums: layers 0 (conv1), 9 (bn1) and 1..3 exist; got %r" % (layer,))
            if len(set(layers)) != len(layers):
                raise NotImplementedError("feature_layer_nums: one hook per layer (got %r)" % (tuple(layers),))
            # forward hooks fire in execution order -- conv1, bn1, layer1.. -- whatever the order of the tuple: that is the order of
            # aux_nets / depth_nets and of the aux columns of the feature rows (models/naive.py:212-241)
            self._hooks = [layer for layer in self.HOOK_ORDER if layer in layers]
            self.early_features = []
            auxs, deps = [], []
            for layer in self._hooks:
                c, h, w = self.HOOK_SHAPES[layer]
                aux, dep = _make_aux(c), _make_depth(h, w)
                if wrap:
                    aux, dep = Replicated(aux), Replicated(dep)
                auxs.append(aux)
                deps.append(dep)
                self.aux_latent_dim += h * w // 4
            if register_heads:
                self.aux_nets = nn.ModuleList(auxs)
                self.depth_nets = nn.ModuleList(deps)
            else:  # TD model: plain lists, invisible to parameters()/state_dict()/.cuda() (time_sensitive.py:102-115)
                self.aux_nets = auxs
                self.depth_nets = deps
            trunk.keep_stem_raw = 0 in self._hooks
        if wrap:
            self.feature_net = Replicated(trunk)
        self._heads_registered = register_heads
        self._arena = None
        self._aux_ops = None
        self._grad_sync = None   # dist.GradSync attached by the data-parallel loop (staged all-reduce under backward)
        self.rollout = False

    # -- helpers --------------------------------------------------------------------------------
    @property
    def trunk(self):
        f = self.feature_net
        return f.module if isinstance(f, Replicated) else f

    def _aux_modules(self, i=0):
        aux, dep = self.aux_nets[i], self.depth_nets[i]
        if isinstance(aux, Replicated):
            aux, dep = aux.module, dep.module
        return aux[0], dep[-2]  # Conv2d, InstanceNorm2d

    def _materialize(self, device):
        """First call on a device: put every parameter into one flat arena and bind gradient views."""
        p0 = next(self.parameters())
        if p0.device != device:
            raise RuntimeError("model parameters are on %s but the batch is on %s: call model.cuda() first" % (p0.device, device))
        if device.type != "cuda":
            raise RuntimeError("the pose models run on the MI355X HIP path only; there is no CPU fallback (got %s tensors)" % device)
        self.trunk.compute_dtype = self.compute_dtype
        self.trunk.ensure_layout()
        if self._arena is None or not self._arena.is_current():
            self._arena = ParamArena(self)
        self._arena.loss_scaler = self.loss_scaler   # FusedAdam unscales / skips through it (amp.py)
        if self.loss_scaler is not None:   # a scaler state restored by FusedAdam.load_state_dict before this model had an arena
            pend = None
            for p in self.parameters():
                if getattr(p, "_rpe_pending_amp", None) is not None:
                    pend = p._rpe_pending_amp
                    p._rpe_pending_amp = None
            if pend is not None:
                self.loss_scaler.load_state_dict(pend)
        if self.aux_nets is not None and (self._aux_ops is None or any(op.conv_w.device != device for op in self._aux_ops)):
            self._aux_ops = []
            for i, layer in enumerate(self._hooks):
                conv, inorm = self._aux_modules(i)
                if not self._heads_registered:
                    for m in (conv, inorm):
                        m.to(device)
                c, h, w = self.HOOK_SHAPES[layer]
                pools = len(self.depth_nets[i].module if isinstance(self.depth_nets[i], Replicated) else self.depth_nets[i]) - 2
                # a conv1 hook sends the stem backward down its unfused path, which takes the bn1 hook's gradient as a dense tensor
                self._aux_ops.append(AuxHeadOp(conv.weight, conv.bias, inorm.weight, inorm.bias, trainable=self._heads_registered, layer=layer,
                                               pools=pools, dense=0 in self._hooks))

    def _anchor(self):
        for p in self.parameters():
            if p.requires_grad:
                return p
        return None

    def _call(self, img, depth, self_measurement):
        self._materialize(img.device)
        # float images are the reference's (…, 3, H, W) tensors; uint8 images are raw (…, Hs, Ws, 3) frames that the trunk
        # crops and normalises on the device
        img = img.contiguous() if img.dtype == torch.uint8 else img.contiguous().float()
        x0bar = None if self_measurement is None else self_measurement.contiguous().float()
        anchor = self._anchor()
        if torch.is_grad_enabled() and self.training and anchor is not None:
            return _ModelFn.apply(self, anchor, img, depth, x0bar)
        with torch.no_grad():
            outs = self._forward_impl(img, depth, x0bar, save=False)
        return outs if len(outs) > 1 else outs[0]

    def _features_fwd(self, img, depth, rows, save):
        """img (B,3,H,W); rows [B, ld] fp32: columns [0,L) <- ResNet latent, [L, L+aux) <- aux head."""
        # the bn1 head of a training forward rides on the engine's stem pass (headops.AuxHeadOp.bind_fused): it is bound before the run
        cols, fused, off = {}, set(), self.latent_dim
        if self.aux_nets is not None:
            for op in self._aux_ops:
                c, h, w = self.HOOK_SHAPES[op.layer]
                cols[id(op)] = rows[:, off:off + h * w // 4]
                off += h * w // 4

        def pre_forward(plan):
            for op in (self._aux_ops or ()) if self.aux_nets is not None else ():
                if op.can_fuse(self.training):
                    op.bind_fused(plan, depth if self.use_depth else None, cols[id(op)], self.use_depth, save=save)
                    fused.add(id(op))

        plan = self.trunk.run(img, rows, self.training, pre_forward=pre_forward)
        if self.aux_nets is not None:
            for op in self._aux_ops:
                if (plan.h, plan.w) != (224, 224) and op.layer != 9:
                    raise ValueError("hooks other than bn1 are sized for 224x224 inputs (as the reference's dummy forward is)")
                if id(op) not in fused:
                    op.fwd(plan, depth if self.use_depth else None, cols[id(op)], self.use_depth, save=save)
        self._plan = plan
        return plan

    def _features_bwd(self, d_rows):
        """d_rows [B, ld >= L + aux] gradient of the fused feature rows."""
        use_early = self.aux_nets is not None and 9 in self._hooks   # (the bn1 hook's gradient enters the stem backward)
        if self.aux_nets is not None:
            off = self.latent_dim
            for op in self._aux_ops:
                c, h, w = self.HOOK_SHAPES[op.layer]
                n = h * w // 4
                op.bwd(d_rows[:, off:off + n])
                off += n
        sync = getattr(self, "_grad_sync", None)
        self._plan.backward(d_rows, use_early, None if sync is None else sync.stage_done)

    @staticmethod
    def _pad_rows(t, cols):
        """(…, cols) tensor -> [rows, cols] view of a zero-padded [rows, pad4(cols)] buffer"""
        t2 = t.reshape(-1, cols)
        out = new_rows(t2.shape[0], cols, t.device)
        ops.copy2d(t2.contiguous().float(), out, cols=cols)
        return out

    def optimizer_stepped(self):
        """The trunk caches compute-dtype copies of its weights; tell it they are stale."""
        self.trunk.weights_changed()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.trunk.weights_changed()
        return r
