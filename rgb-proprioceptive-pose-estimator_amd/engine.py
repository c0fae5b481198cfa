"""Host side of the ResNet-50 trunk: a torchvision-shaped parameter container (same attribute
names and state_dict keys as the module util/model_utils.py:136-141 builds) whose compute is the
native launch plan in csrc/engine.hip.  The nn.Conv2d / nn.BatchNorm2d / nn.Linear objects below
only HOLD parameters and buffers (names, shapes, initialisers, state_dict, .cuda()); their
forward() is never called -- calling this module without the HIP library or on CPU tensors raises.
"""
import ctypes

import torch
import torch.nn as nn

from . import ops
from ._lib import ResizePlan, lib
from .util.data_utils import crop_origin, pil_bilinear_tables, resized_hw

# util/model_utils.py:130-136: the bottleneck members 50 / 101 / 152 and the BasicBlock member 18 (its "32" is no torchvision model; 34 runs on the same plan)
_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
_BASIC = (18, 34)
_PLANES_STRIDES = ((64, 1), (128, 2), (256, 2), (512, 2))


class _Shortcut(nn.Sequential):
    pass


class _BottleneckParams(nn.Module):
    def __init__(self, inplanes, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if project:
            self.downsample = _Shortcut(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, *a, **k):
        raise RuntimeError("parameter container only: the trunk runs as one native plan (ResNet50Trunk.run)")


class _BasicBlockParams(nn.Module):
    """torchvision BasicBlock (ResNet-18 / 34): conv3x3(stride) - bn - relu - conv3x3 - bn, + identity / projection, relu"""
    def __init__(self, inplanes, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        if project:
            self.downsample = _Shortcut(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, *a, **k):
        raise RuntimeError("parameter container only: the trunk runs as one native plan (ResNet50Trunk.run)")


class ResNet50Trunk(nn.Module):
    """ResNet v1.5 parameters (depth 50 by default; 101 / 152, BasicBlock 18 / 34) + native forward/backward plan.

    compute_dtype: torch.bfloat16 (default; fp32 accumulate, fp32 master weights) or torch.float32
    (exact-fp32 MFMA path used for the 1e-4 parity bar).
    """

    def __init__(self, num_outputs=1000, compute_dtype=torch.bfloat16, depth=50):
        super().__init__()
        if depth not in _BLOCKS:
            raise NotImplementedError("no native MI355X launch plan for resnet%d (18, 34, 50, 101, 152 have one)" % depth)
        self.depth = depth
        self.expansion = 1 if depth in _BASIC else 4
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, ((planes, stride), n) in enumerate(zip(_PLANES_STRIDES, _BLOCKS[depth]), start=1):
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                if depth in _BASIC:
                    blocks.append(_BasicBlockParams(inpl, planes, s, b == 0 and (s != 1 or inpl != planes)))
                else:
                    blocks.append(_BottleneckParams(inpl, planes, s, b == 0))
                inpl = planes * self.expansion
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
        self.fc = nn.Linear(512 * self.expansion, num_outputs)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.compute_dtype = compute_dtype
        # device-side preprocessing of uint8 frames (the reference's transform, util/data_utils.py:48-54)
        self.resize_to = 256   # Resize(256): frames whose shorter side differs are resampled on the device (Pillow's bilinear)
        self._resize_plans = {}
        self.crop_hw = (224, 224)
        self.norm_mean = (0.485, 0.456, 0.406)
        self.norm_std = (0.229, 0.224, 0.225)
        self._plans = {}      # (B, H, W, dtype, latent, device) -> _Plan, most recently used last
        self.max_plans = 3    # each plan owns a multi-GB workspace: the train plan plus the latest eval / rollout shapes stay
        self._active = None
        self.keep_stem_raw = False   # a forward hook sits on conv1: inference keeps conv1's raw output too (models/_core.py)
        self._wver = 0        # bumped whenever the fp32 masters or the BN running statistics may have moved

    # -- plumbing ---------------------------------------------------------------------------------
    def _ordered(self):
        """(conv, bn) pairs and the parameter / buffer tables in the engine's order."""
        pairs = [(self.conv1, self.bn1)]
        for li in range(1, 5):
            for blk in getattr(self, "layer%d" % li):
                pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
                if hasattr(blk, "conv3"):
                    pairs.append((blk.conv3, blk.bn3))
                if hasattr(blk, "downsample"):
                    pairs.append((blk.downsample[0], blk.downsample[1]))
        params = []
        for conv, bn in pairs:
            params += [conv.weight, bn.weight, bn.bias]
        params += [self.fc.weight, self.fc.bias]
        running = []
        for _, bn in pairs:
            running += [bn.running_mean, bn.running_var]
        nbt = [bn.num_batches_tracked for _, bn in pairs]
        return pairs, params, running, nbt

    def ensure_layout(self):
        """Conv weights (except the stem) are kept in channels_last storage = [Co][kh][kw][Ci], the
        layout the implicit-GEMM kernels read; values/shape/state_dict are unchanged."""
        pairs, _, _, _ = self._ordered()
        for i, (conv, _) in enumerate(pairs):
            w = conv.weight
            if i == 0:
                if not w.data.is_contiguous():
                    w.data = w.data.contiguous()
            elif not w.data.permute(0, 2, 3, 1).is_contiguous():
                w.data = w.data.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)

    def _plan(self, batch, h, w):
        key = (batch, h, w, self.compute_dtype, self.fc.out_features, self.fc.weight.device)
        plan = self._plans.pop(key, None)
        if plan is None:
            while len(self._plans) >= max(1, self.max_plans):   # evict the least recently used plan (frees its workspace)
                old_key = next(iter(self._plans))
                if self._plans[old_key] is self._active:
                    self._active = None
                del self._plans[old_key]
            plan = _Plan(self, batch, h, w)
        self._plans[key] = plan
        plan.rebind_if_needed()
        return plan

    def body_frozen(self):
        """True when no parameter of the ResNet body takes a gradient (import_resnet froze it: feature_extract and use_pretrained):
        the backward then computes the replaced fc's gradient only."""
        fc = {id(self.fc.weight), id(self.fc.bias)}
        return not any(p.requires_grad for p in self.parameters() if id(p) not in fc)

    def weights_changed(self):
        """Call after an optimizer step or load_state_dict: every plan's cached weight copies (compute-dtype copies for
        training, BN-folded copies for inference) are stale.  Training forwards call it themselves."""
        self._wver += 1

    # -- compute ----------------------------------------------------------------------------------
    def run(self, img, features, training, pre_forward=None):
        """img (B,3,H,W) fp32 device tensor; features: fp32 2-D device tensor whose first `latent`
        columns receive the ResNet output.  Returns the plan (early feature / backward handle).
        pre_forward(plan): called once the plan is chosen and before the engine's forward is enqueued (the bn1 aux head hands the
        engine its parameters and output columns there: headops.AuxHeadOp.bind_fused)."""
        if not img.is_cuda:
            raise RuntimeError("ResNet50Trunk needs device tensors: the HIP path has no CPU fallback")
        frames = img.dtype == torch.uint8  # raw simulator frames (B, Hs, Ws, 3): cropped + normalised on the device
        if frames:
            b, hs, ws, c = img.shape
            h, w = self.crop_hw
            assert c == 3 and img.is_contiguous()
        else:
            b, c, h, w = img.shape
            assert c == 3 and img.dtype == torch.float32 and img.is_contiguous()
        plan = self._plan(b, h, w)
        s = ops._stream()
        if training:
            # the optimizer moved the fp32 masters since the last step -> refresh the compute-dtype / transposed copies
            # (one packing launch, ~0.2 GB of traffic)
            lib.rpe_resnet50_pack_weights(plan.handle, s)
        elif plan.wver != self._wver:
            # inference on a plan whose BN-folded copies were built before the masters / running statistics last moved --
            # possibly through ANOTHER plan (train() runs train and val phases at different batch sizes): rebuild them
            lib.rpe_resnet50_weights_changed(plan.handle)
        plan.wver = self._wver
        lib.rpe_resnet50_set_stem_raw(plan.handle, int(self.keep_stem_raw))
        if pre_forward is not None:
            pre_forward(plan)
        if frames:
            F3 = ctypes.c_float * 3
            hr, wr = resized_hw(hs, ws, self.resize_to)
            if min(hr, wr) < min(h, w) or hr < h or wr < w:
                raise ValueError("frames of %dx%d resize to %dx%d, smaller than the %dx%d crop" % (hs, ws, hr, wr, h, w))
            if (hr, wr) == (hs, ws) and (hs - h) % 2 == 0 and (ws - w) % 2 == 0:
                lib.rpe_resnet50_forward_u8(plan.handle, ops._p(img), hs, ws, F3(*self.norm_mean), F3(*self.norm_std), ops._p(features),
                                            features.stride(0), int(training), s)
            else:
                rs = self._resize_plan(b, hs, ws, hr, wr, h, w, img.device)
                lib.rpe_resnet50_forward_u8_resized(plan.handle, ops._p(img), hs, ws, ctypes.byref(rs["plan"]), F3(*self.norm_mean), F3(*self.norm_std),
                                                    ops._p(features), features.stride(0), int(training), s)
        else:
            lib.rpe_resnet50_forward(plan.handle, ops._p(img), ops._p(features), features.stride(0), int(training), s)
        self._active = plan
        if training:
            self._wver += 1   # running statistics moved now; the masters will with the optimizer step that follows
        return plan

    def _resize_plan(self, b, hs, ws, hr, wr, h, w, device):
        """device tap tables + intermediate buffer of the Pillow-exact resize for one frame geometry (built once, kept)"""
        key = (b, hs, ws, hr, wr, h, w, device)
        rs = self._resize_plans.get(key)
        if rs is None:
            top, left = crop_origin(hr, wr, h, w)
            tabs = {}
            for name, (i, o) in (("x", (ws, wr)), ("y", (hs, hr))):
                if i != o:
                    bounds, kk = pil_bilinear_tables(i, o)
                    tabs[name] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).contiguous().to(device), kk.shape[1])
            tmp = torch.empty(b * hs * wr * 3, dtype=torch.uint8, device=device) if "x" in tabs else None
            xb, xk, ksx = tabs.get("x", (None, None, 0))
            yb, yk, ksy = tabs.get("y", (None, None, 0))
            ptr = lambda t: None if t is None else t.data_ptr()
            plan = ResizePlan(hr, wr, top, left, ksx, ksy, ptr(xb), ptr(xk), ptr(yb), ptr(yk), ptr(tmp))
            rs = {"plan": plan, "keep": (xb, xk, yb, yk, tmp)}
            self._resize_plans[key] = rs
        return rs

    def forward(self, img):
        """Inference-style call (no autograd): returns the latent features (B, latent)."""
        lat = self.fc.out_features
        out = torch.empty((img.shape[0], ops.pad4(lat)), dtype=torch.float32, device=img.device)
        self.run(img, out, self.training)
        return out[:, :lat]


class _Plan:
    """One native engine object + its workspace for a fixed (batch, H, W, dtype)."""

    def __init__(self, trunk, batch, h, w):
        self.trunk = trunk
        self.batch, self.h, self.w = batch, h, w
        self.dtype = trunk.compute_dtype
        hp = ctypes.c_void_p()
        lib.rpe_resnet_create(ctypes.byref(hp), trunk.depth, batch, h, w, ops.dtype_code(self.dtype), trunk.fc.out_features)
        self.handle = hp
        nbytes = lib.rpe_resnet50_workspace_bytes(hp)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=trunk.fc.weight.device)
        self._ptr_sig = None
        self.wver = None      # trunk._wver the engine's cached weight copies were built at
        self.grad_views = None

    def __del__(self):
        try:
            if self.handle:
                lib.rpe_resnet50_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def rebind_if_needed(self):
        t = self.trunk
        t.ensure_layout()
        _, params, running, nbt = t._ordered()
        grads = [getattr(p, "_rpe_grad", None) for p in params]
        sig = tuple(p.data_ptr() for p in params) + tuple(0 if g is None else g.data_ptr() for g in grads) + tuple(r.data_ptr() for r in running)
        if sig == self._ptr_sig:
            return
        n = len(params)
        PA = ctypes.c_void_p * n
        pa = PA(*[p.data_ptr() for p in params])
        have_grads = all(g is not None for g in grads)
        ga = PA(*[g.data_ptr() for g in grads]) if have_grads else None
        RA = ctypes.c_void_p * len(running)
        ra = RA(*[r.data_ptr() for r in running])
        NA = ctypes.c_void_p * len(nbt)
        na = NA(*[x.data_ptr() for x in nbt])
        base = self.workspace.data_ptr()
        off = (-base) % 256
        lib.rpe_resnet50_bind(self.handle, ctypes.c_void_p(base + off), self.workspace.numel() - off, pa, ga, ra, na)
        self._ptr_sig = sig
        self.wver = None

    # early feature relu(bn1(conv1 x)) as an NHWC tensor aliasing the workspace
    def early_feature(self):
        ptr = lib.rpe_resnet50_early_feature(self.handle)
        if not ptr:
            raise RuntimeError("the last forward computed the bn1 aux head inside the stem's apply + pool pass and did not write relu(bn1(conv1 x)) "
                               "(rpe_resnet50_set_aux_head); set RPE_NO_AUX_FUSE=1 to keep the tensor")
        return self._alias(ptr, (self.batch, self.h // 2, self.w // 2, 64))

    def hooked_feature(self, layer):
        """The tensor a forward hook on `layer` sees (models/naive.py:201-211): 0 conv1's raw output, 9 bn1 (after the in-place
        ReLU), 1..3 the output of layerN -- NHWC, compute dtype, aliasing the workspace."""
        if layer == 9:
            return self.early_feature()
        if layer == 0:
            return self.tensor("conv1.y").view(self.batch, self.h // 2, self.w // 2, 64)
        t = self.tensor("layer%d.%d.%s.a" % (layer, _BLOCKS[self.trunk.depth][layer - 1] - 1, "conv2" if self.trunk.depth in _BASIC else "conv3"))
        s = 2 << layer   # layer1: 1/4 of the image, layer2: 1/8, layer3: 1/16
        return t.view(self.batch, self.h // s, self.w // s, t.shape[1])

    def set_hook_grad(self, layer, dense):
        """dense gradient of hooked_feature(layer) (layer 0..3), added by the next backward; keep it alive until then"""
        lib.rpe_resnet50_set_hook_grad(self.handle, layer, ops._p(dense))

    def early_grad(self):
        ptr = lib.rpe_resnet50_early_grad(self.handle)
        return self._alias(ptr, (self.batch, self.h // 2, self.w // 2, 64))

    def _alias(self, ptr, shape):
        base = self.workspace.data_ptr()
        esz = 4 if self.dtype == torch.float32 else 2
        n = 1
        for s in shape:
            n *= s
        off = ptr - base
        return self.workspace[off:off + n * esz].view(self.dtype).view(shape)

    def set_aux_grad(self, d_cols, depth_feat, idx, conv_w):
        """Hand the aux head's gradient to the next backward in compact form (the stem backward gathers it on the fly); the
        tensors must stay alive until that backward has been enqueued."""
        lib.rpe_resnet50_set_aux_grad(self.handle, ops._p(d_cols), d_cols.stride(0), ops._p(depth_feat), ops._p(idx), ops._p(conv_w))

    def tensor(self, name):
        ptr, rows, ch = ctypes.c_void_p(), ctypes.c_long(), ctypes.c_int()
        lib.rpe_resnet50_tensor(self.handle, name.encode(), ctypes.byref(ptr), ctypes.byref(rows), ctypes.byref(ch))
        return self._alias(ptr.value, (rows.value, ch.value))

    # (stage name, index into the per-stage block counts), in backward order

    def backward(self, d_features, use_d_early, stage_done=None):
        """stage_done(name): optional callback fired when the gradients of a stage ("fc", "layer4" .. "layer1", "stem") are
        complete in stream order -- the data-parallel path starts their all-reduce there, under the rest of the backward."""
        s = ops._stream()
        if self.trunk.body_frozen():
            # feature_extract and use_pretrained (util/model_utils.py:110-113,137 of the reference): nothing flows into the body
            lib.rpe_resnet50_backward_frozen(self.handle, ops._p(d_features), d_features.stride(0), s)
            if stage_done is not None:
                for name in ("fc", "layer4", "layer3", "layer2", "layer1", "stem"):
                    stage_done(name)
            return
        if stage_done is None:
            lib.rpe_resnet50_backward(self.handle, ops._p(d_features), d_features.stride(0), int(use_d_early), s)
            return
        lib.rpe_resnet50_backward_begin(self.handle, ops._p(d_features), d_features.stride(0), s)
        stage_done("fc")
        counts = _BLOCKS[self.trunk.depth]
        for li in (4, 3, 2, 1):
            name, nblocks = "layer%d" % li, counts[li - 1]
            lib.rpe_resnet50_backward_blocks(self.handle, nblocks, 1, s)
            stage_done(name)
        lib.rpe_resnet50_backward_end(self.handle, int(use_d_early), s)
        stage_done("stem")
