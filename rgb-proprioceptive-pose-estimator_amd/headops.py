"""Forward/backward of the head layers (Linear(+ReLU), LSTM, aux head, depth head) as explicit
op objects over the C ABI.  Each op's fwd() keeps what its bwd() needs; bwd() WRITES parameter
gradients into the parameters' arena gradient views (p._rpe_grad) and returns the input gradient.
Everything is fp32; row buffers are [rows, pad4(cols)] with zero padding columns so every GEMM
operand is 16-byte aligned.
"""
import os

import torch

from . import ops
from ._lib import lib

pad4 = ops.pad4


def new_rows(rows, cols, device, zero=True):
    """[rows, cols] view of a fresh [rows, pad4(cols)] fp32 buffer (padding columns are zero)."""
    buf = (torch.zeros if zero else torch.empty)((rows, pad4(cols)), dtype=torch.float32, device=device)
    return buf[:, :cols]


def _grad_of(p):
    g = getattr(p, "_rpe_grad", None)
    if g is None:
        raise RuntimeError("parameter has no arena gradient view (model not materialised on the device)")
    return g


class LinearOp:
    """y = x W^T + b (+ReLU).  replaces nn.Linear / F.relu(fc(x)) (models/naive.py:343-345)."""

    def __init__(self, weight, bias, relu=False):
        self.weight, self.bias, self.relu = weight, bias, relu
        self._wpad = None

    def _w(self):
        w = self.weight.data
        n, k = w.shape
        if k % 4 == 0 and w.is_contiguous():
            return w
        if self._wpad is None or self._wpad.device != w.device:
            self._wpad = torch.zeros((n, pad4(k)), dtype=torch.float32, device=w.device)
        ops.copy2d(w, self._wpad, cols=k)
        return self._wpad

    def fwd(self, x, save=True, out=None, addend=None):
        """x: [M, K] view (row stride multiple of 4, columns K..pad4(K) zero)."""
        n, k = self.weight.shape
        if x.shape[0] <= ops.LINEAR_ROWS_MAX and self.weight.data.is_contiguous():
            # few rows (a rollout frame): the per-column kernel reads the weight in place, whatever its row alignment -- no padded
            # copy (15 MB per frame for the first fusion layer); an inference output's padding columns are never read
            if out is None:
                out = new_rows(x.shape[0], n, x.device, zero=save)
            ops.linear_fwd(x, self.weight.data, None if self.bias is None else self.bias.data, relu=self.relu, addend=addend, out=out, n=n, k=k)
            if save:
                self.x, self.y = x, out
            return out
        w = self._w()
        if out is None:
            out = new_rows(x.shape[0], n, x.device)
        ops.linear_fwd(x, w, None if self.bias is None else self.bias.data, relu=self.relu, addend=addend, out=out, n=n, k=pad4(k))
        if save:
            self.x, self.y = x, out
        return out

    def bwd(self, dy, need_dx=True, accumulate=False):
        """dy: [M, N] view with zero padding columns.  Returns dx [M, K] (or None)."""
        n, k = self.weight.shape
        x, y = self.x, self.y
        m = x.shape[0]
        if self.relu:
            # y and dy are rows of contiguous [M, pad4(N)] buffers: mask the padded buffer (padding stays zero)
            if y.stride(0) != pad4(n) or dy.stride(0) != pad4(n):
                raise RuntimeError("LinearOp.bwd: ReLU layers need [M, pad4(N)] row buffers")
            ypad = y.as_strided((m, pad4(n)), (pad4(n), 1), y.storage_offset())
            dypad = dy.as_strided((m, pad4(n)), (pad4(n), 1), dy.storage_offset())
            dy = ops.relu_bwd(ypad, dypad)[:, :n]
        gw = _grad_of(self.weight)
        if k % 4 == 0 and gw.is_contiguous():
            ops.linear_wgrad(dy, x, gw, n=n, k=k, overwrite=not accumulate)   # (overwriting: no zero fill in front)
        else:
            tmp = torch.zeros((n, pad4(k)), dtype=torch.float32, device=dy.device)
            ops.linear_wgrad(dy, x, tmp, n=n, k=pad4(k))
            if accumulate:
                gw.add_(tmp[:, :k])
            else:
                ops.copy2d(tmp, gw, cols=k)
        if self.bias is not None:
            ops.colsum(dy, cols=n, out=_grad_of(self.bias), accumulate=accumulate)
        if not need_dx:
            return None
        wt = ops.transpose_f32(self.weight.data, ldo=pad4(n))  # [K, pad4(N)]
        dx = new_rows(m, k, dy.device)
        ops.linear_fwd(dy, wt, None, relu=False, out=dx, n=k, k=pad4(n))
        return dx


class LSTMOp:
    """Single-layer nn.LSTM over (S, N, I), torch gate order (models/time_sensitive.py:212,235,501,759,768).
    The input GEMM runs once over all S*N rows; each step is one recurrent GEMM + one fused cell kernel."""

    def __init__(self, w_ih, w_hh, b_ih, b_hh):
        self.w_ih, self.w_hh, self.b_ih, self.b_hh = w_ih, w_hh, b_ih, b_hh
        self.in_lin = LinearOp(w_ih, None, relu=False)
        self.hidden = w_hh.shape[1]
        if self.hidden % 4:
            raise ValueError("LSTM hidden size must be a multiple of 4 (16-byte rows)")

    def fwd(self, x, S, N, h0=None, c0=None, save=True):
        """x: [S*N, I] row view.  Returns (h_all [S*N, H] contiguous, (h_T, c_T))."""
        H = self.hidden
        dev = x.device
        xg = self.in_lin.fwd(x, save=save)  # [S*N, 4H], no bias
        gates = torch.empty((S, N, 4 * H), dtype=torch.float32, device=dev)
        hs = torch.empty((S, N, H), dtype=torch.float32, device=dev)
        cs = torch.empty((S, N, H), dtype=torch.float32, device=dev)
        whh = self.w_hh.data
        s = ops._stream()
        for t in range(S):
            h_prev = h0 if t == 0 else hs[t - 1]
            c_prev = c0 if t == 0 else cs[t - 1]
            xg_t = xg[t * N:(t + 1) * N]
            if h_prev is None:
                gates[t].copy_(xg_t)
            else:
                ops.linear_fwd(h_prev, whh, None, addend=xg_t, out=gates[t], n=4 * H, k=H)
            lib.rpe_lstm_cell_fwd(ops._p(gates[t]), ops._p(self.b_ih.data), ops._p(self.b_hh.data), ops._p(c_prev), ops._p(cs[t]), ops._p(hs[t]),
                                  N, H, s)
        if save:
            self.S, self.N, self.gates, self.hs, self.cs, self.h0, self.c0 = S, N, gates, hs, cs, h0, c0
        return hs.view(S * N, H), (hs[S - 1], cs[S - 1])

    def bwd(self, dh_all, need_dx=True):
        """dh_all: [S*N, H] gradient wrt every step's hidden output.  Returns dx [S*N, I] (or None)."""
        S, N, H = self.S, self.N, self.hidden
        dev = dh_all.device
        dgates = torch.empty((S, N, 4 * H), dtype=torch.float32, device=dev)
        dc = torch.zeros((N, H), dtype=torch.float32, device=dev)
        whh_t = ops.transpose_f32(self.w_hh.data, ldo=4 * H)  # [H, 4H]
        s = ops._stream()
        dh_all = dh_all.reshape(S, N, H).contiguous()
        dh = dh_all[S - 1]
        for t in reversed(range(S)):
            c_prev = self.c0 if t == 0 else self.cs[t - 1]
            lib.rpe_lstm_cell_bwd(ops._p(self.gates[t]), ops._p(c_prev), ops._p(self.cs[t]), ops._p(dh), ops._p(dc), ops._p(dgates[t]), N, H, s)
            if t > 0:
                # gradient of h_{t-1}: its own output gradient + the recurrent term, in one launch (the GEMM's addend); H % 4 == 0,
                # so the row buffer has no padding columns to zero
                dh = new_rows(N, H, dev, zero=False)
                ops.linear_fwd(dgates[t], whh_t, None, out=dh, addend=dh_all[t - 1], n=H, k=4 * H)
        dg = dgates.view(S * N, 4 * H)
        g_hh = _grad_of(self.w_hh)
        g_hh.zero_()
        if self.h0 is not None:
            ops.linear_wgrad(dgates[0], self.h0, g_hh, n=4 * H, k=H)
        if S > 1:
            ops.linear_wgrad(dg[N:], self.hs[:S - 1].reshape((S - 1) * N, H), g_hh, n=4 * H, k=H)
        ops.colsum(dg, cols=4 * H, out=_grad_of(self.b_ih))
        _grad_of(self.b_hh).copy_(_grad_of(self.b_ih))
        return self.in_lin.bwd(dg, need_dx=need_dx)


class AuxHeadOp:
    """Conv2d(C->1,1x1)+MaxPool2d(2)+Flatten on a hooked feature map, optionally multiplied by the depth feature
    (models/naive.py:223-240,318-330).  layer 9 (bn1, the hook every reference script uses) runs the 64-channel kernels whose
    gradient the fused stem backward gathers in compact form; the other hooks (conv1, layer1..3) and `dense` mode (a conv1 hook
    forces the unfused stem backward) run the general kernels with a dense feature gradient."""

    def __init__(self, conv_w, conv_b, in_w, in_b, trainable=True, layer=9, pools=2, dense=False):
        self.conv_w, self.conv_b, self.in_w, self.in_b = conv_w, conv_b, in_w, in_b
        self.trainable = trainable
        self.layer, self.pools, self.dense = layer, pools, dense

    def can_fuse(self, training):
        """The bn1 head of a TRAINING forward can ride on the stem's apply + pool pass (round 4): the engine computes it and never
        writes relu(bn1(conv1 x)).  Not in dense mode (a conv1 hook / RPE_STEM_UNFUSED need the tensor), not in inference."""
        return (training and self.layer == 9 and not self.dense and os.environ.get("RPE_STEM_UNFUSED") is None
                and os.environ.get("RPE_NO_AUX_FUSE") is None)

    def _depth_feature(self, depth, b, h, w, n, dev, use_depth):
        feat = xhat = None
        if use_depth:
            if depth is None or depth.shape[-2:] != (2 * h, 2 * w):
                raise ValueError("use_depth=True needs a (B,1,H,W) depth batch matching the image size")
            depth = depth.reshape(b, 2 * h, 2 * w).contiguous()
            feat = torch.empty((b, n), dtype=torch.float32, device=dev)
            xhat = torch.empty((b, n), dtype=torch.float32, device=dev)
            lib.rpe_depth_head_fwd(ops._p(depth), ops._p(self.in_w.data), ops._p(self.in_b.data), ops._p(feat), ops._p(xhat), b, 2 * h, 2 * w, ops._stream())
        return feat, xhat

    def bind_fused(self, plan, depth, out_cols, use_depth, save=True):
        """Called BEFORE the trunk's forward (ResNet50Trunk.run(pre_forward=...)): the depth feature is computed here, the engine is
        handed the head's parameters and output columns and computes the head inside its stem pass (rpe_resnet50_set_aux_head)."""
        b, h, w = plan.batch, plan.h // 2, plan.w // 2
        n = (h // 2) * (w // 2)
        dev = out_cols.device
        raw = torch.empty((b, n), dtype=torch.float32, device=dev)
        idx = torch.empty((b, n), dtype=torch.uint8, device=dev)
        feat, xhat = self._depth_feature(depth, b, h, w, n, dev, use_depth)
        lib.rpe_resnet50_set_aux_head(plan.handle, ops._p(self.conv_w.data), ops._p(self.conv_b.data), ops._p(feat), ops._p(out_cols), out_cols.stride(0),
                                      ops._p(raw), ops._p(idx))
        self._bound = (raw, idx, feat)   # alive until the forward has been enqueued
        if save:
            self.saved = (plan, raw, idx, feat, xhat, b, h, w, n)
        self.fused = True
        return out_cols

    def fwd(self, plan, depth, out_cols, use_depth, save=True):
        """out_cols: [B, (H/2)*(W/2)] column slice of the fused feature rows."""
        if self.layer != 9:
            return self._fwd_general(plan, depth, out_cols, use_depth, save)
        self.fused = False
        a1 = plan.early_feature()
        b, h, w, _ = a1.shape
        n = (h // 2) * (w // 2)
        dev = a1.device
        raw = torch.empty((b, n), dtype=torch.float32, device=dev)
        idx = torch.empty((b, n), dtype=torch.uint8, device=dev)
        s = ops._stream()
        feat, xhat = self._depth_feature(depth, b, h, w, n, dev, use_depth)
        lib.rpe_aux_head_fwd(ops.dtype_code(a1), ops._p(a1), ops._p(self.conv_w.data), ops._p(self.conv_b.data), ops._p(feat), ops._p(out_cols),
                             out_cols.stride(0), ops._p(raw), ops._p(idx), b, h, w, s)
        if save:
            self.saved = (plan, raw, idx, feat, xhat, b, h, w, n)
        return out_cols

    def _fwd_general(self, plan, depth, out_cols, use_depth, save):
        x = plan.hooked_feature(self.layer)
        b, h, w, c = x.shape
        n = (h // 2) * (w // 2)
        dev = x.device
        raw = torch.empty((b, n), dtype=torch.float32, device=dev)
        idx = torch.empty((b, n), dtype=torch.uint8, device=dev)
        feat = xhat = None
        s = ops._stream()
        if use_depth:
            if depth is None or depth.dim() < 2:
                raise ValueError("use_depth=True needs a (B,1,H,W) depth batch")
            hd, wd = depth.shape[-2:]
            if (hd >> self.pools, wd >> self.pools) != (h // 2, w // 2):
                raise ValueError("depth batch of %dx%d does not pool to the %dx%d aux feature of layer %d" % (hd, wd, h // 2, w // 2, self.layer))
            depth = depth.reshape(b, hd, wd).contiguous().float()
            feat = torch.empty((b, n), dtype=torch.float32, device=dev)
            xhat = torch.empty((b, n), dtype=torch.float32, device=dev)
            lib.rpe_depth_head_fwd_pools(ops._p(depth), ops._p(self.in_w.data), ops._p(self.in_b.data), ops._p(feat), ops._p(xhat), b, hd, wd, self.pools, s)
        lib.rpe_aux_head_fwd_c(ops.dtype_code(x), ops._p(x), c, ops._p(self.conv_w.data), ops._p(self.conv_b.data), ops._p(feat), ops._p(out_cols),
                               out_cols.stride(0), ops._p(raw), ops._p(idx), b, h, w, s)
        if save:
            self.saved = (plan, raw, idx, feat, xhat, b, h, w, n)
        return out_cols

    def _bwd_general(self, d_cols):
        plan, raw, idx, feat, xhat, b, h, w, n = self.saved
        x = plan.hooked_feature(self.layer)
        c = x.shape[3]
        dev = x.device
        if self.trainable:
            gw, gb = _grad_of(self.conv_w), _grad_of(self.conv_b)
            gw.zero_(), gb.zero_()
        else:
            gw = torch.zeros(c, dtype=torch.float32, device=dev)
            gb = torch.zeros(1, dtype=torch.float32, device=dev)
        d_feat = torch.empty((b, n), dtype=torch.float32, device=dev) if feat is not None else None
        d_x = torch.empty_like(x)                     # dense: zero except the winner pixel of every 2x2 window
        s = ops._stream()
        lib.rpe_aux_head_bwd_c(ops.dtype_code(x), ops._p(d_cols), d_cols.stride(0), ops._p(x), c, ops._p(self.conv_w.data), ops._p(feat), ops._p(raw),
                               ops._p(idx), ops._p(d_x), ops._p(gw), ops._p(gb), ops._p(d_feat), b, h, w, s)
        self._keep = d_x                              # alive until the trunk backward has been enqueued
        plan.set_hook_grad(self.layer, d_x)
        if feat is not None and self.trainable:
            giw, gib = _grad_of(self.in_w), _grad_of(self.in_b)
            giw.zero_(), gib.zero_()
            lib.rpe_depth_head_bwd(ops._p(d_feat), ops._p(xhat), b * n, ops._p(giw), ops._p(gib), s)

    def bwd(self, d_cols):
        """d_cols: [B, (H/2)*(W/2)] column slice of the feature-row gradient.  Fills the trunk's early-feature
        gradient buffer and the head's parameter gradients."""
        if self.layer != 9:
            return self._bwd_general(d_cols)
        plan, raw, idx, feat, xhat, b, h, w, n = self.saved
        fused = getattr(self, "fused", False)   # the forward rode on the stem pass: a1 does not exist, the engine recomputes it from y
        a1 = None if fused else plan.early_feature()
        # The gradient of a1 is NOT materialised (a 411 MB tensor at 256 images, zero in 3 of 4 pixels): the trunk's fused stem
        # backward gathers it from (d_cols, depth feature, winner index, conv weight).  RPE_STEM_UNFUSED=1: dense form.
        dense = self.dense or os.environ.get("RPE_STEM_UNFUSED") is not None
        d_a1 = plan.early_grad() if dense else None
        dev = d_cols.device
        if self.trainable:
            gw, gb = _grad_of(self.conv_w), _grad_of(self.conv_b)
            gw.zero_(), gb.zero_()
        else:  # TD model quirk: heads are not registered parameters (time_sensitive.py:102-115)
            gw = torch.zeros(64, dtype=torch.float32, device=dev)
            gb = torch.zeros(1, dtype=torch.float32, device=dev)
        d_feat = torch.empty((b, n), dtype=torch.float32, device=dev) if feat is not None else None
        s = ops._stream()
        # parameter gradients through per-block partial sums added in a fixed order (bitwise reproducible, and faster than 65 global
        # atomics per block on the same 65 addresses)
        code = ops.dtype_code(plan.dtype)
        nws = lib.rpe_aux_head_bwd_workspace_floats(code, b, h, w)
        if getattr(self, "_part", None) is None or self._part.numel() < nws or self._part.device != dev:
            self._part = torch.empty(nws, dtype=torch.float32, device=dev)
        if fused:
            if dense:
                raise RuntimeError("AuxHeadOp: a fused forward cannot be followed by the dense backward")
            lib.rpe_resnet50_aux_head_bwd(plan.handle, ops._p(d_cols), d_cols.stride(0), ops._p(self.conv_w.data), ops._p(feat), ops._p(raw), ops._p(idx),
                                          ops._p(gw), ops._p(gb), ops._p(d_feat), ops._p(self._part), self._part.numel(), s)
        else:
            lib.rpe_aux_head_bwd_det(code, ops._p(d_cols), d_cols.stride(0), ops._p(a1), ops._p(self.conv_w.data), ops._p(feat), ops._p(raw),
                                     ops._p(idx), ops._p(d_a1), ops._p(gw), ops._p(gb), ops._p(d_feat), b, h, w, ops._p(self._part), self._part.numel(), s)
        if not dense:
            self._keep = (d_cols, feat, idx)   # alive until the trunk backward has been enqueued
            plan.set_aux_grad(d_cols, feat, idx, self.conv_w.data)
        if feat is not None and self.trainable:
            giw, gib = _grad_of(self.in_w), _grad_of(self.in_b)
            giw.zero_(), gib.zero_()
            lib.rpe_depth_head_bwd(ops._p(d_feat), ops._p(xhat), b * n, ops._p(giw), ops._p(gib), s)
