"""Thin tensor-level wrappers over the C ABI (include/rpe_hip.h).

PyTorch is used for device memory and streams only: every function here allocates its
outputs with torch.empty on the input's device, passes raw device pointers plus the
current HIP stream to librpe_hip.so, and returns the output tensors.  Activations of the
conv trunk are NHWC tensors ([B, H, W, C], contiguous) in fp32 or bf16.
"""
import ctypes
import os

import torch

from ._lib import RPE_BF16, RPE_F16, RPE_F32, BnBwdEpilogue, ConvDesc, lib

_DT = {torch.float32: RPE_F32, torch.bfloat16: RPE_BF16, torch.float16: RPE_F16}


def dtype_code(t):
    try:
        return _DT[t if isinstance(t, torch.dtype) else t.dtype]
    except KeyError:
        raise TypeError("unsupported compute dtype %r (fp32, bf16 or fp16)" % (t,))


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("librpe_hip ops need device tensors (got a CPU tensor): the HIP path has no CPU fallback")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name):
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def pad4(n):
    return (n + 3) // 4 * 4


def conv_desc(x_shape, out_c, k, stride, pad):
    b, h, w, c = x_shape
    return ConvDesc(b, h, w, c, out_c, k, k, stride, pad)


def conv_out_hw(d):
    return (d.in_h + 2 * d.pad - d.kh) // d.stride + 1, (d.in_w + 2 * d.pad - d.kw) // d.stride + 1


def stats_tiles(rows):
    return (rows + 127) // 128


def conv2d_fwd(x, w_krsc, stride, pad, want_stats=False):
    """x [B,H,W,Ci], w_krsc [Co,kh,kw,Ci] (same dtype) -> y [B,Ho,Wo,Co] (+ stats partials [tiles,2,Co] fp32)."""
    _chk(x, "x"), _chk(w_krsc, "w")
    co, k = w_krsc.shape[0], w_krsc.shape[1]
    d = conv_desc(x.shape, co, k, stride, pad)
    ho, wo = conv_out_hw(d)
    y = torch.empty((x.shape[0], ho, wo, co), dtype=x.dtype, device=x.device)
    st = None
    if want_stats:
        st = torch.empty((lib.rpe_conv2d_fwd_stats_tiles(ctypes.byref(d), dtype_code(x)), 2, co), dtype=torch.float32, device=x.device)
    lib.rpe_conv2d_fwd(ctypes.byref(d), dtype_code(x), _p(x), _p(w_krsc), _p(y), _p(st), _stream())
    return (y, st) if want_stats else y


def conv2d_fwd_affine(x, w_krsc, stride, pad, bias, addend=None, relu=True, split_k=True):
    """Inference form: relu(conv(x, w) + bias (+ addend)) in one launch (BatchNorm folded: w = weight * scale, bias = shift)."""
    _chk(x, "x"), _chk(w_krsc, "w")
    co, k = w_krsc.shape[0], w_krsc.shape[1]
    d = conv_desc(x.shape, co, k, stride, pad)
    ho, wo = conv_out_hw(d)
    out = torch.empty((x.shape[0], ho, wo, co), dtype=x.dtype, device=x.device)
    need = lib.rpe_conv2d_fwd_affine_workspace_bytes(ctypes.byref(d), dtype_code(x)) if split_k else 0
    if need > 0:   # few output tiles, long K (a rollout frame): split-K through a workspace
        ws = scratch(need, x.device)
        lib.rpe_conv2d_fwd_affine_ws(ctypes.byref(d), dtype_code(x), _p(x), _p(w_krsc), _p(out), _p(bias), _p(addend), int(relu), _p(ws), ws.numel(),
                                     _stream())
    else:
        lib.rpe_conv2d_fwd_affine(ctypes.byref(d), dtype_code(x), _p(x), _p(w_krsc), _p(out), _p(bias), _p(addend), int(relu), _stream())
    return out


def conv2d_dgrad(dy, w_crsk, x_shape, stride, pad, addend=None):
    """dy [B,Ho,Wo,Co], w_crsk [Ci,kh,kw,Co] -> dx [B,H,W,Ci] (+ addend)."""
    _chk(dy, "dy"), _chk(w_crsk, "w")
    ci, k, co = w_crsk.shape[0], w_crsk.shape[1], w_crsk.shape[3]
    d = conv_desc(x_shape, co, k, stride, pad)
    dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    lib.rpe_conv2d_dgrad(ctypes.byref(d), dtype_code(dy), _p(dy), _p(w_crsk), _p(dx), _p(addend), _stream())
    return dx


def conv2d_dgrad_bn(dy, w_crsk, x_shape, stride, pad, y, mean, invstd, a_out=None, scale=None, shift=None, addend=None, a_mask=None):
    """Data gradient with the producing layer's ReLU mask and BN-backward partial sums fused into the epilogue.
    Returns (dz [B,H,W,Ci], stats partials [tiles,2,Ci])."""
    _chk(dy, "dy"), _chk(w_crsk, "w")
    if y is not None:   # (y = None with a_mask: the producing layer's raw output was never written -- only sum dz is emitted)
        _chk(y, "y")
    ci, k, co = w_crsk.shape[0], w_crsk.shape[1], w_crsk.shape[3]
    d = conv_desc(x_shape, co, k, stride, pad)
    dz = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    st = torch.empty((lib.rpe_conv2d_dgrad_stats_tiles(ctypes.byref(d), dtype_code(dy)), 2, ci), dtype=torch.float32, device=dy.device)
    ep = BnBwdEpilogue(*(None if t is None else t.data_ptr() for t in (y, a_out, mean, invstd, scale, shift, st, a_mask)))
    lib.rpe_conv2d_dgrad_bn(ctypes.byref(d), dtype_code(dy), _p(dy), _p(w_crsk), _p(dz), _p(addend), ctypes.byref(ep), _stream())
    return dz, st


def conv1x1_dgrad_bn_t(dy, w_crsk, x_shape, a_prev, a_mask, addend=None):
    """The fused 1x1 data gradient with the producing layer's packed ReLU mask (sum dz only) that also leaves T = dz^T a_prev behind.
    Returns (dz [B,H,W,Ci], stats partials [tiles,2,Ci], T [Ci, P] fp32)."""
    _chk(dy, "dy"), _chk(w_crsk, "w"), _chk(a_prev, "a_prev")
    ci, co = w_crsk.shape[0], w_crsk.shape[3]
    pch = a_prev.shape[-1]
    d = conv_desc(x_shape, co, 1, 1, 0)
    dz = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    st = torch.empty((lib.rpe_conv2d_dgrad_stats_tiles(ctypes.byref(d), dtype_code(dy)), 2, ci), dtype=torch.float32, device=dy.device)
    t_out = torch.empty((ci, pch), dtype=torch.float32, device=dy.device)
    ws = scratch(lib.rpe_conv1x1_dgrad_bn_t_workspace_bytes(ctypes.byref(d), pch), dy.device)
    ep = BnBwdEpilogue(*(None if t is None else t.data_ptr() for t in (None, None, None, None, None, None, st, a_mask)))
    lib.rpe_conv1x1_dgrad_bn_t(ctypes.byref(d), dtype_code(dy), _p(dy), _p(w_crsk), _p(dz), _p(addend), ctypes.byref(ep), _p(a_prev), pch, _p(t_out),
                               _p(ws), ws.numel(), _stream())
    return dz, st, t_out


def bn_backward_from_dz(dz, y, mean, invstd, gamma, stats_part):
    c = y.shape[-1]
    dev = y.device
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    dy = torch.empty_like(y)
    c1c2 = torch.empty(2 * c, dtype=torch.float32, device=dev)
    dpart = torch.zeros(256 * 2 * c + 64, dtype=torch.float64, device=dev)  # RPE_BN_DPART_DOUBLES(c); counters start at zero
    lib.rpe_bn_backward_from_dz(dtype_code(y), _p(dz), _p(y), _p(mean), _p(invstd), _p(gamma), _p(stats_part), stats_part.shape[0], _p(dgamma),
                                _p(dbeta), _p(dy), y.numel() // c, c, _p(c1c2), _p(dpart), _stream())
    return dy, dgamma, dbeta


_SCRATCH = {}


def last_kernel_name():
    """Symbol (with its configuration) of the kernel the last implicit-GEMM / BN entry point launched from this thread."""
    return lib.rpe_last_kernel_name().decode()


def set_walk_direction(mode):
    """Walk direction of this thread's next launches of the direction-aware kernels (rpe_set_walk_direction): 0 every XCD walks its
    share of the row tiles upwards (default), 1 downwards, 2 alternating launch by launch.  Scheduling only: results do not depend on it."""
    lib.rpe_set_walk_direction(int(mode))


def scratch(nbytes, device):
    """Per-device workspace for the deterministic weight-gradient calls (per-workgroup fp32 tiles, summed in a fixed order).
    One buffer that only ever grows: consecutive calls on one stream are ordered, so they can share it -- and its address stays
    fixed, which a captured hipGraph needs.  Use one stream per device for these ops (the models do)."""
    key = (device.type, device.index)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def bn_backward_coeffs(stats_part, rows):
    """per-tile partial sums [tiles, 2, C] of (dz, dz*xhat) -> (dgamma, dbeta, c1c2 [2, C] = their means)"""
    tiles, _, c = stats_part.shape
    dev = stats_part.device
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    c1c2 = torch.empty((2, c), dtype=torch.float32, device=dev)
    dpart = torch.zeros(256 * 2 * c + 64, dtype=torch.float64, device=dev)
    lib.rpe_bn_backward_coeffs(_p(stats_part), tiles, c, rows, _p(dgamma), _p(dbeta), _p(c1c2), _p(dpart), _stream())
    return dgamma, dbeta, c1c2


def bn_backward_coeffs_t(stats_part, rows, dzt_a, w, mean, invstd):
    """y3-free block: partial sums [tiles, 2, C] whose first half is sum dz, T = dz^T a_in [C, Ci] fp32 and the compute-dtype weight
    [C, Ci] -> (dgamma, dbeta, c1c2): sum dz*xhat = invstd (rowdot(T, w) - mean sum dz)"""
    tiles, _, c = stats_part.shape
    ci = w.shape[1]
    dev = stats_part.device
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    c1c2 = torch.empty((2, c), dtype=torch.float32, device=dev)
    dpart = torch.zeros(256 * 2 * c + 64, dtype=torch.float64, device=dev)
    lib.rpe_bn_backward_coeffs_t(dtype_code(w), _p(stats_part), tiles, c, rows, _p(_chk(dzt_a, "dzt_a")), _p(_chk(w, "w")), ci, _p(mean), _p(invstd), _p(dgamma),
                                 _p(dbeta), _p(c1c2), _p(dpart), _stream())
    return dgamma, dbeta, c1c2


def conv1x1_wgrad_combine(dzt_a, gram_buf, w_master, gamma, invstd, mean, c1c2):
    """dW [Co, Ci] fp32 of a y3-free block's conv3: A o (T - c1 s1^T) + C' o (W S - mean s1^T), S / s1 from the forward's Gram buffer"""
    co, ci = dzt_a.shape
    d = conv_desc((1, 1, 1, ci), co, 1, 1, 0)
    dw = torch.empty((co, ci), dtype=torch.float32, device=dzt_a.device)
    lib.rpe_conv1x1_wgrad_combine(ctypes.byref(d), _p(_chk(dzt_a, "dzt_a")), _p(gram_buf), _p(_chk(w_master, "w")), _p(gamma), _p(invstd), _p(mean), _p(c1c2),
                                  _p(dw), _stream())
    return dw


def bn_backward_apply_dz(dz, y, mean, invstd, gamma, c1c2):
    dy = torch.empty_like(y)
    c = y.shape[-1]
    lib.rpe_bn_backward_apply_dz(dtype_code(y), _p(dz), _p(y), _p(mean), _p(invstd), _p(gamma), _p(c1c2), _p(dy), y.numel() // c, c, _stream())
    return dy


def bn_bwd_fold_conv1x1(w_fwd, w_dgrad, gamma, invstd, mean, c1c2):
    """w_fwd [Co, Ci], w_dgrad [Ci, Co] (compute dtype) -> (w_kcat [Ci, Co + Ci], bias [Ci] fp32): the BN backward of the conv's
    output folded into its data gradient (see include/rpe_hip.h)."""
    co, ci = w_fwd.shape
    dev = w_fwd.device
    wk = torch.empty((ci, co + ci), dtype=w_fwd.dtype, device=dev)
    bias = torch.empty(ci, dtype=torch.float32, device=dev)
    ws = scratch(lib.rpe_bn_bwd_fold_scratch_bytes(dtype_code(w_fwd), co, ci), dev)
    lib.rpe_bn_bwd_fold_conv1x1(dtype_code(w_fwd), co, ci, _p(_chk(w_fwd, "w_fwd")), _p(_chk(w_dgrad, "w_dgrad")), _p(gamma), _p(invstd), _p(mean), _p(c1c2),
                                _p(wk), _p(bias), _p(ws), ws.numel(), _stream())
    return wk, bias


def bn_bwd_fold_y_conv1x1(w_dgrad, gamma, invstd, mean, c1c2):
    """w_dgrad [Ci, Co] (compute dtype) -> (w_kcat [Ci, 2 Co], bias [Ci] fp32): the backward of the BatchNorm BEHIND a 1x1 conv whose
    output is narrow (a Bottleneck's conv1), folded into its data gradient with y itself as the second K-concatenated operand."""
    ci, co = w_dgrad.shape
    dev = w_dgrad.device
    wk = torch.empty((ci, 2 * co), dtype=w_dgrad.dtype, device=dev)
    bias = torch.empty(ci, dtype=torch.float32, device=dev)
    lib.rpe_bn_bwd_fold_y_conv1x1(dtype_code(w_dgrad), co, ci, _p(_chk(w_dgrad, "w_dgrad")), _p(gamma), _p(invstd), _p(mean), _p(c1c2), _p(wk), _p(bias), _stream())
    return wk, bias


def conv1x1_dgrad_kcat_y(dz, y, w_kcat, bias, ci, addend=None, bn=None):
    """dz, y [B,H,W,Co] -> dx [B,H,W,Ci] = [dz | y] w_kcat^T + bias (+ addend).  bn: optional dict(y, mean, invstd, a_out / a_mask, scale,
    shift) of the layer BEHIND dx (fused ReLU mask + BN-backward partial sums); then returns (dz_in, stats)."""
    _chk(dz, "dz"), _chk(y, "y")
    b, h, w, co = dz.shape
    d = conv_desc((b, h, w, ci), co, 1, 1, 0)
    dx = torch.empty((b, h, w, ci), dtype=dz.dtype, device=dz.device)
    if bn is None:
        lib.rpe_conv1x1_dgrad_kcat_y(ctypes.byref(d), dtype_code(dz), _p(dz), _p(y), _p(w_kcat), _p(bias), _p(dx), _p(addend), None, _stream())
        return dx
    st = torch.empty((lib.rpe_conv2d_dgrad_stats_tiles(ctypes.byref(d), dtype_code(dz)), 2, ci), dtype=torch.float32, device=dz.device)
    ep = BnBwdEpilogue(*(None if t is None else t.data_ptr() for t in (bn["y"], bn.get("a_out"), bn["mean"], bn["invstd"], bn.get("scale"), bn.get("shift"), st,
                                                                      bn.get("a_mask"))))
    lib.rpe_conv1x1_dgrad_kcat_y(ctypes.byref(d), dtype_code(dz), _p(dz), _p(y), _p(w_kcat), _p(bias), _p(dx), _p(addend), ctypes.byref(ep), _stream())
    return dx, st


def conv1x1_wgrad_folded_y(dz, y, x, gamma, invstd, mean, c1c2):
    """dW [Co, Ci] fp32 of a 1x1 conv with the backward of the BatchNorm behind it folded in: reads dz, y (both [.., Co]) and x ([.., Ci])."""
    _chk(dz, "dz"), _chk(y, "y"), _chk(x, "x")
    b, h, w, co = dz.shape
    ci = x.shape[3]
    d = conv_desc((b, h, w, ci), co, 1, 1, 0)
    dw = torch.empty((co, ci), dtype=torch.float32, device=dz.device)
    ws = scratch(lib.rpe_conv1x1_wgrad_folded_y_scratch_bytes(ctypes.byref(d), dtype_code(dz)), dz.device)
    lib.rpe_conv1x1_wgrad_folded_y(ctypes.byref(d), dtype_code(dz), _p(dz), _p(y), _p(x), _p(gamma), _p(invstd), _p(mean), _p(c1c2), _p(dw), _p(ws), ws.numel(),
                                   _stream())
    return dw


def conv1x1_wgrad_folded(dz, a_in, w_master, gamma, invstd, mean, c1c2):
    """dW [Co, Ci] fp32 of a 1x1 conv with its output BN's backward folded in: reads dz and the conv input only."""
    _chk(dz, "dz"), _chk(a_in, "a_in")
    b, h, w, co = dz.shape
    ci = a_in.shape[3]
    d = conv_desc((b, h, w, ci), co, 1, 1, 0)
    dw = torch.empty((co, ci), dtype=torch.float32, device=dz.device)
    ws = scratch(lib.rpe_conv1x1_wgrad_folded_scratch_bytes(ctypes.byref(d), dtype_code(dz)), dz.device)
    lib.rpe_conv1x1_wgrad_folded(ctypes.byref(d), dtype_code(dz), _p(dz), _p(a_in), _p(_chk(w_master, "w")), _p(gamma), _p(invstd), _p(mean), _p(c1c2), _p(dw),
                                 _p(ws), ws.numel(), _stream())
    return dw


def conv1x1_dgrad_kcat(dz, a_in, w_kcat, bias, bn=None):
    """dz [B,H,W,Co], a_in [B,H,W,Ci] -> dx [B,H,W,Ci].  bn: optional dict(y, mean, invstd, scale, shift, a_out, a_mask) of the
    layer BEHIND a_in (fused ReLU mask + BN-backward partial sums, as conv2d_dgrad_bn); then returns (dz_in, stats)."""
    _chk(dz, "dz"), _chk(a_in, "a_in")
    b, h, w, co = dz.shape
    ci = a_in.shape[3]
    d = conv_desc((b, h, w, ci), co, 1, 1, 0)
    dx = torch.empty_like(a_in)
    if bn is None:
        lib.rpe_conv1x1_dgrad_kcat(ctypes.byref(d), dtype_code(dz), _p(dz), _p(a_in), _p(w_kcat), _p(bias), _p(dx), None, _stream())
        return dx
    st = torch.empty((lib.rpe_conv2d_dgrad_stats_tiles(ctypes.byref(d), dtype_code(dz)), 2, ci), dtype=torch.float32, device=dz.device)
    ep = BnBwdEpilogue(*(None if t is None else t.data_ptr() for t in (bn["y"], bn.get("a_out"), bn["mean"], bn["invstd"], bn.get("scale"), bn.get("shift"), st,
                                                                      bn.get("a_mask"))))
    lib.rpe_conv1x1_dgrad_kcat(ctypes.byref(d), dtype_code(dz), _p(dz), _p(a_in), _p(w_kcat), _p(bias), _p(dx), ctypes.byref(ep), _stream())
    return dx, st


def conv2d_wgrad(x, dy, k, stride, pad, deterministic=True):
    """-> dw [Co,kh,kw,Ci] fp32.  deterministic: slab + fixed-order sum (bitwise reproducible); else fp32 atomics."""
    _chk(x, "x"), _chk(dy, "dy")
    co = dy.shape[3]
    d = conv_desc(x.shape, co, k, stride, pad)
    if deterministic:
        dw = torch.empty((co, k, k, x.shape[3]), dtype=torch.float32, device=x.device)
        ws = scratch(lib.rpe_conv2d_wgrad_workspace_bytes(ctypes.byref(d), dtype_code(x)), x.device)
        lib.rpe_conv2d_wgrad_det(ctypes.byref(d), dtype_code(x), _p(x), _p(dy), _p(dw), _p(ws), ws.numel(), _stream())
        return dw
    dw = torch.zeros((co, k, k, x.shape[3]), dtype=torch.float32, device=x.device)
    lib.rpe_conv2d_wgrad(ctypes.byref(d), dtype_code(x), _p(x), _p(dy), _p(dw), _stream())
    return dw


def pack_conv_weight(w_krsc_f32, dtype):
    """fp32 [Co,kh,kw,Ci] -> (forward copy in `dtype`, dgrad copy [Ci,kh,kw,Co] in `dtype`)."""
    co, kh, kw, ci = w_krsc_f32.shape
    wf = torch.empty((co, kh, kw, ci), dtype=dtype, device=w_krsc_f32.device)
    wd = torch.empty((ci, kh, kw, co), dtype=dtype, device=w_krsc_f32.device)
    lib.rpe_pack_conv_weight(dtype_code(dtype), _p(_chk(w_krsc_f32, "w")), _p(wf), _p(wd), co, kh, kw, ci, _stream())
    return wf, wd


STEM_PAD = 3   # RPE_STEM_PAD: the staged image is zero-bordered, [B, H + 6, W + 6, 4]


def stage_image(img_nchw, dtype):
    """(B, 3, H, W) fp32 -> the zero-bordered NHWC4 image [B, H + 6, W + 6, 4] the stem kernels read (interior written by the kernel)"""
    b, c, h, w = img_nchw.shape
    assert c == 3 and img_nchw.dtype == torch.float32
    out = torch.zeros((b, h + 2 * STEM_PAD, w + 2 * STEM_PAD, 4), dtype=dtype, device=img_nchw.device)
    lib.rpe_stage_image_nhwc4(dtype_code(dtype), _p(_chk(img_nchw, "img")), _p(out), b, h, w, _stream())
    return out


def pack_stem_weight(w_oihw, dtype):
    out = torch.empty((64, 8, 8, 4), dtype=dtype, device=w_oihw.device)
    lib.rpe_pack_stem_weight(dtype_code(dtype), _p(_chk(w_oihw, "w")), None, _p(out), _stream())
    return out


def stem_conv_fwd(x4, w_packed, want_stats=False):
    b, h, w, _ = x4.shape
    h, w = h - 2 * STEM_PAD, w - 2 * STEM_PAD
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    y = torch.empty((b, ho, wo, 64), dtype=x4.dtype, device=x4.device)
    st = torch.empty((stats_tiles(b * ho * wo), 2, 64), dtype=torch.float32, device=x4.device) if want_stats else None
    lib.rpe_stem_conv_fwd(dtype_code(x4), _p(x4), _p(w_packed), _p(y), _p(st), b, h, w, _stream())
    return (y, st) if want_stats else y


def stem_conv_wgrad(x4, dy):
    b, h, w, _ = x4.shape
    h, w = h - 2 * STEM_PAD, w - 2 * STEM_PAD
    dwp = torch.zeros((64, 8, 8, 4), dtype=torch.float32, device=x4.device)
    ws = scratch(lib.rpe_stem_conv_wgrad_workspace_bytes(dtype_code(x4), b, h, w), x4.device)
    lib.rpe_stem_conv_wgrad_det(dtype_code(x4), _p(x4), _p(dy), _p(dwp), b, h, w, _p(ws), ws.numel(), _stream())
    dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x4.device)
    lib.rpe_unpack_stem_grad(_p(dwp), _p(dw), _stream())
    return dw


def bn_finalize(part, count, gamma, beta, running_mean=None, running_var=None, num_batches=None, momentum=0.1, eps=1e-5):
    tiles, _, c = part.shape
    dev = part.device
    scale, shift, mean, invstd = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
    dpart = torch.zeros(256 * 2 * c + 64, dtype=torch.float64, device=dev)  # RPE_BN_DPART_DOUBLES(c); counters start at zero
    lib.rpe_bn_finalize(_p(part), tiles, c, count, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(num_batches),
                        momentum, eps, _p(scale), _p(shift), _p(mean), _p(invstd), _p(dpart), _stream())
    return scale, shift, mean, invstd


def bn_apply(y, scale, shift, residual=None, relu=True):
    out = torch.empty_like(y)
    c = y.shape[-1]
    lib.rpe_bn_apply(dtype_code(y), _p(_chk(y, "y")), _p(residual), _p(out), _p(scale), _p(shift), y.numel() // c, c, int(relu), _stream())
    return out


def bn_apply_mask(y, scale, shift, residual=None):
    """relu(y*scale + shift (+ residual)) and its packed ReLU mask (1 byte per 8 channels); 16-bit element types only."""
    out = torch.empty_like(y)
    c = y.shape[-1]
    mask = torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device)
    lib.rpe_bn_apply_mask(dtype_code(y), _p(_chk(y, "y")), _p(residual), _p(out), _p(scale), _p(shift), y.numel() // c, c, _p(mask), _stream())
    return out, mask


def bn_apply_res_bn(y, scale, shift, res_y, res_scale, res_shift, relu=True, want_mask=False):
    """[relu](y*scale + shift + res_y*res_scale + res_shift): the residual is a raw conv output under its own BatchNorm (the
    projection shortcut), applied on the fly; want_mask: also the packed ReLU mask (16-bit element types)."""
    out = torch.empty_like(y)
    c = y.shape[-1]
    mask = torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device) if want_mask else None
    lib.rpe_bn_apply_res_bn(dtype_code(y), _p(_chk(y, "y")), _p(res_y), _p(res_scale), _p(res_shift), _p(out), _p(scale), _p(shift), y.numel() // c, c,
                            int(relu), _p(mask), _stream())
    return (out, mask) if want_mask else out


def gram(x):
    """x [..., C] (NHWC activations) -> (x^T x [C, C], colsum(x) [C], the whole buffer) fp32, one launch + fixed-order slab sum"""
    c = x.shape[-1]
    m = x.numel() // c
    code = dtype_code(x)
    ones_row = lib.rpe_gram_ones_row(c)
    out = torch.empty((ones_row + 1, c), dtype=torch.float32, device=x.device)
    ws = scratch(lib.rpe_gram_workspace_bytes(code, m, c), x.device)
    lib.rpe_gram(code, _p(_chk(x, "x")), m, c, _p(out), _p(ws), ws.numel(), _stream())
    return out[:c], out[ones_row], out


def bn_apply_gram(y, scale, shift):
    """relu(y * scale + shift) and the Gram buffer of the result (as ops.gram) in one pass: (out, S [C, C], s1 [C], buffer)"""
    c = y.shape[-1]
    m = y.numel() // c
    code = dtype_code(y)
    ones_row = lib.rpe_gram_ones_row(c)
    buf = torch.zeros((ones_row + 1, c), dtype=torch.float32, device=y.device)
    out = torch.empty_like(y)
    ws = scratch(lib.rpe_bn_apply_gram_workspace_bytes(code, m, c), y.device)
    lib.rpe_bn_apply_gram(code, _p(_chk(y, "y")), _p(out), _p(scale), _p(shift), m, c, _p(buf), _p(ws), ws.numel(), _stream())
    return out, buf[:c], buf[ones_row], buf


def bn_stats_from_gram(w, gram_buf, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """BatchNorm statistics of y = x w^T from the Gram buffer of x (ops.gram): (scale, shift, mean, invstd)"""
    co, ci = w.shape
    dev = w.device
    scale, shift, mean, invstd = (torch.empty(co, dtype=torch.float32, device=dev) for _ in range(4))
    lib.rpe_bn_stats_from_gram(dtype_code(w), _p(_chk(w, "w")), co, ci, _p(gram_buf), lib.rpe_gram_ones_row(ci), int(count), _p(gamma), _p(beta),
                               _p(running_mean), _p(running_var), None, momentum, eps, _p(scale), _p(shift), _p(mean), _p(invstd), _stream())
    return scale, shift, mean, invstd


def conv1x1_fwd_bn(x, w, scale, shift, residual=None, res_scale=None, res_shift=None, want_y=False):
    """relu((x w^T) * scale + shift + residual [* res_scale + res_shift]) with the packed ReLU mask, one launch (16-bit types)"""
    co, ci = w.shape
    d = conv_desc(x.shape, co, 1, 1, 0)
    out = torch.empty(tuple(x.shape[:-1]) + (co,), dtype=x.dtype, device=x.device)
    y = torch.empty_like(out) if want_y else None
    mask = torch.empty(out.numel() // 8, dtype=torch.uint8, device=x.device)
    lib.rpe_conv1x1_fwd_bn(ctypes.byref(d), dtype_code(x), _p(_chk(x, "x")), _p(_chk(w, "w")), _p(out), _p(y), _p(scale), _p(shift), _p(residual),
                           _p(res_scale), _p(res_shift), _p(mask), _stream())
    return out, mask, y


def bn_backward(dA, a_out, y, mean, invstd, gamma, want_dz=False):
    c = y.shape[-1]
    dev = y.device
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    dy = torch.empty_like(y)
    dz = torch.empty_like(y) if want_dz else None
    part = torch.empty(2 * 1024 * c, dtype=torch.float32, device=dev)
    c1c2 = torch.empty(2 * c, dtype=torch.float32, device=dev)
    dpart = torch.zeros(256 * 2 * c + 64, dtype=torch.float64, device=dev)  # RPE_BN_DPART_DOUBLES(c); counters start at zero
    lib.rpe_bn_backward(dtype_code(y), _p(dA), _p(a_out), _p(y), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dy), _p(dz),
                        y.numel() // c, c, _p(part), part.numel(), _p(c1c2), _p(dpart), _stream())
    return dy, dgamma, dbeta, dz


def maxpool_fwd(x):
    b, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((b, ho, wo, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=x.device)
    lib.rpe_maxpool3x3s2_fwd(dtype_code(x), _p(_chk(x, "x")), _p(out), _p(idx), b, h, w, c, _stream())
    return out, idx


def bn_apply_maxpool(y, scale, shift):
    """a = relu(y * scale + shift), pooled = maxpool3x3s2(a), winner taps -- one pass over y [B,H,W,C] (H, W even)"""
    b, h, w, c = y.shape
    a = torch.empty_like(y)
    out = torch.empty((b, h // 2, w // 2, c), dtype=y.dtype, device=y.device)
    idx = torch.empty((b, h // 2, w // 2, c), dtype=torch.uint8, device=y.device)
    lib.rpe_bn_apply_maxpool3x3s2(dtype_code(y), _p(_chk(y, "y")), _p(scale), _p(shift), _p(a), _p(out), _p(idx), b, h, w, c, _stream())
    return a, out, idx


def maxpool_bwd(dout, idx, x_shape, addend=None):
    b, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), dtype=dout.dtype, device=dout.device)
    lib.rpe_maxpool3x3s2_bwd(dtype_code(dout), _p(dout), _p(idx), _p(addend), _p(dx), b, h, w, c, _stream())
    return dx


def avgpool_fwd(x):
    b, h, w, c = x.shape
    out = torch.empty((b, c), dtype=torch.float32, device=x.device)
    lib.rpe_avgpool_fwd(dtype_code(x), _p(_chk(x, "x")), _p(out), b, h * w, c, _stream())
    return out


def avgpool_bwd(dout, x_shape, dtype):
    b, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), dtype=dtype, device=dout.device)
    lib.rpe_avgpool_bwd(dtype_code(dtype), _p(dout), _p(dx), b, h * w, c, _stream())
    return dx


# rows up to which rpe_linear_fwd (fp32) runs one workgroup per output column and accepts any row stride / alignment
# (csrc/conv_api.hip: linear_rows_kernel); 0 when switched off
LINEAR_ROWS_MAX = 0 if os.environ.get("RPE_NO_LINEAR_ROWS") else 8
LINEAR_SPLIT_K = not os.environ.get("RPE_NO_LINEAR_SPLITK")   # split-K for Linear launches with few tiles and long K


def linear_fwd(x, w, bias=None, relu=False, addend=None, out=None, n=None, k=None):
    """fp32 (or bf16) y[M, N] = x[M, :K] @ w[:N, :K]^T.  x, w, out, addend are 2-D with arbitrary (chunk-multiple) row strides."""
    m = x.shape[0]
    k = x.shape[1] if k is None else k
    n = w.shape[0] if n is None else n
    if out is None:
        out = torch.zeros((m, pad4(n)), dtype=x.dtype, device=x.device)[:, :n]
    need = lib.rpe_linear_fwd_workspace_bytes(dtype_code(x), m, n, k) if LINEAR_SPLIT_K else 0
    if need > 0:   # few output tiles, long K: split-K through the shared workspace (fixed-order sums)
        ws = scratch(need, x.device)
        lib.rpe_linear_fwd_ws(dtype_code(x), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), m, n, k, int(relu),
                              _p(addend), 0 if addend is None else addend.stride(0), _p(ws), ws.numel(), _stream())
        return out
    lib.rpe_linear_fwd(dtype_code(x), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), m, n, k, int(relu),
                       _p(addend), 0 if addend is None else addend.stride(0), _stream())
    return out


def linear_wgrad(dy, x, dw, n=None, k=None, deterministic=True, overwrite=False):
    """dw[:N, :K] (fp32) += dy[M, :N]^T @ x[M, :K].  deterministic: per-workgroup slabs summed in a fixed order, then ONE add
    into dw per element (bitwise reproducible); else fp32 atomics.  overwrite (deterministic form only): dw = ... instead of +=."""
    m = x.shape[0]
    n = dy.shape[1] if n is None else n
    k = x.shape[1] if k is None else k
    if deterministic:
        ws = scratch(lib.rpe_linear_wgrad_workspace_bytes(dtype_code(x), m, n, k), x.device)
        lib.rpe_linear_wgrad_det(dtype_code(x), _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw.stride(0), m, n, k, 0 if overwrite else 1, _p(ws),
                                 ws.numel(), _stream())
        return dw
    if overwrite:
        dw.zero_()
    lib.rpe_linear_wgrad(dtype_code(x), _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw.stride(0), m, n, k, _stream())
    return dw


def transpose_f32(w, ldo=None):
    rows, cols = w.shape
    ldo = pad4(rows) if ldo is None else ldo
    out = torch.empty((cols, ldo), dtype=torch.float32, device=w.device)
    lib.rpe_transpose_f32(_p(w), _p(out), rows, cols, w.stride(0), ldo, _stream())
    return out


def relu_bwd(out, dy):
    dx = torch.empty_like(dy)
    assert out.is_contiguous() and dy.is_contiguous()
    lib.rpe_relu_bwd(_p(out), _p(dy), _p(dx), dy.numel(), _stream())
    return dx


def colsum(x, cols=None, out=None, accumulate=False):
    rows = x.shape[0]
    cols = x.shape[1] if cols is None else cols
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    lib.rpe_colsum(_p(x), rows, cols, x.stride(0), _p(out), int(accumulate), _stream())
    return out


def copy2d(src, dst, cols=None):
    cols = src.shape[1] if cols is None else cols
    lib.rpe_copy2d(_p(src), src.stride(0), _p(dst), dst.stride(0), src.shape[0], cols, _stream())
    return dst


def pose_loss(pred, truth, metric, mode, scale, alpha, eps, want_grad=True):
    """pred/truth (..., 7) fp32 contiguous -> (out3 [loss, val_pos, val_ori], grad or None)."""
    n = pred.numel() // 7
    out3 = torch.empty(3, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    lib.rpe_pose_loss(_p(_chk(pred, "pred")), _p(_chk(truth, "truth")), n, metric, mode, scale, alpha, eps, _p(out3), _p(grad), _stream())
    return out3, grad


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    lib.rpe_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, _stream())
