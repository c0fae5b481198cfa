"""PyTorch dispatcher registration of the HIP operators: `torch.ops.rpe.*`.

The product boundary is the C ABI (include/rpe_hip.h); the model classes reach it through ctypes (ops.py, engine.py).  This module is
the INNER face SURVEY.md section 8(b) describes for a maintainer who wants the operators as PyTorch custom ops: every entry is
registered with the dispatcher for the CUDA (= HIP on ROCm) device type ONLY -- a call with CPU tensors fails in the dispatcher
("no implementation for device cpu"), there is no fallback -- with a fake (meta) implementation giving shapes / dtypes, and the two
differentiable ones (`rpe::conv2d`, `rpe::pose_distance_loss`) carry their backward formulas, built from the same HIP launches.

Each operator names the torch call of the reference's hot path it stands for (models/naive.py:316 runs them through torchvision's
ResNet; models/losses.py:114-128 is the loss; util/learn_utils.py:152-184 the step):

    rpe::conv2d_fwd / conv2d_dgrad / conv2d_wgrad   F.conv2d and its two gradients, NHWC activations, [Co,kh,kw,Ci] weights
    rpe::conv2d                                     the three as ONE differentiable op
    rpe::bn_apply                                   BatchNorm's affine map (+ residual) (+ ReLU) on an NHWC tensor
    rpe::linear_fwd                                 F.linear (+ ReLU)
    rpe::pose_loss / pose_distance_loss             PoseDistanceLoss (raw three-value form / differentiable scalar)
    rpe::adam_step                                  torch.optim.Adam's update of one flat fp32 tensor, in place

Importing this module needs torch only; the HIP library is loaded on the first call (ops.py), so the schemas can be inspected on a
machine without a GPU (tests/test_host_cpu.py).
"""
from typing import List, Optional, Tuple

import torch

__all__ = ["NAMES"]

_NS = "rpe"
NAMES = ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv2d", "bn_apply", "linear_fwd", "pose_loss", "pose_distance_loss", "adam_step")


def _ops():
    from . import ops   # (loads librpe_hip.so: raises ImportError with the build instructions when it is missing)
    return ops


def _out_hw(h, w, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


# ---- convolution (NHWC, compute dtype fp32 / bf16 / fp16) ---------------------------------------------------------------------
@torch.library.custom_op(_NS + "::conv2d_fwd", mutates_args=(), device_types="cuda")
def conv2d_fwd(x: torch.Tensor, w_krsc: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    return _ops().conv2d_fwd(x, w_krsc, stride, pad)


@conv2d_fwd.register_fake
def _(x, w_krsc, stride, pad):
    ho, wo = _out_hw(x.shape[1], x.shape[2], w_krsc.shape[1], stride, pad)
    return x.new_empty((x.shape[0], ho, wo, w_krsc.shape[0]))


@torch.library.custom_op(_NS + "::conv2d_dgrad", mutates_args=(), device_types="cuda")
def conv2d_dgrad(dy: torch.Tensor, w_crsk: torch.Tensor, x_shape: List[int], stride: int, pad: int) -> torch.Tensor:
    return _ops().conv2d_dgrad(dy, w_crsk, tuple(x_shape), stride, pad)


@conv2d_dgrad.register_fake
def _(dy, w_crsk, x_shape, stride, pad):
    return dy.new_empty(tuple(x_shape))


@torch.library.custom_op(_NS + "::conv2d_wgrad", mutates_args=(), device_types="cuda")
def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    """-> fp32 [Co, k, k, Ci], the deterministic form (per-workgroup slabs summed in a fixed order)"""
    return _ops().conv2d_wgrad(x, dy, k, stride, pad, deterministic=True)


@conv2d_wgrad.register_fake
def _(x, dy, k, stride, pad):
    return x.new_empty((dy.shape[3], k, k, x.shape[3]), dtype=torch.float32)


@torch.library.custom_op(_NS + "::conv2d", mutates_args=(), device_types="cuda")
def conv2d(x: torch.Tensor, w_krsc: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    """differentiable: x [B,H,W,Ci] and w [Co,kh,kw,Ci] of one compute dtype; d/dw comes back in that dtype (rounded from the fp32 sum)"""
    return _ops().conv2d_fwd(x, w_krsc, stride, pad)


@conv2d.register_fake
def _(x, w_krsc, stride, pad):
    ho, wo = _out_hw(x.shape[1], x.shape[2], w_krsc.shape[1], stride, pad)
    return x.new_empty((x.shape[0], ho, wo, w_krsc.shape[0]))


def _conv2d_setup(ctx, inputs, output):
    x, w, stride, pad = inputs
    ctx.save_for_backward(x, w)
    ctx.stride, ctx.pad = stride, pad


def _conv2d_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = dw = None
    if ctx.needs_input_grad[0]:
        w_crsk = w.permute(3, 1, 2, 0).contiguous()   # [Ci, kh, kw, Co]: the data gradient's weight layout
        dx = torch.ops.rpe.conv2d_dgrad(dy, w_crsk, list(x.shape), ctx.stride, ctx.pad)
    if ctx.needs_input_grad[1]:
        dw = torch.ops.rpe.conv2d_wgrad(x, dy, w.shape[1], ctx.stride, ctx.pad).to(w.dtype)
    return dx, dw, None, None


conv2d.register_autograd(_conv2d_backward, setup_context=_conv2d_setup)


# ---- BatchNorm affine map, Linear ---------------------------------------------------------------------------------------------
@torch.library.custom_op(_NS + "::bn_apply", mutates_args=(), device_types="cuda")
def bn_apply(y: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, residual: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    return _ops().bn_apply(y, scale, shift, residual, relu)


@bn_apply.register_fake
def _(y, scale, shift, residual, relu):
    return torch.empty_like(y)


@torch.library.custom_op(_NS + "::linear_fwd", mutates_args=(), device_types="cuda")
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    return _ops().linear_fwd(x, w, bias, relu).contiguous()


@linear_fwd.register_fake
def _(x, w, bias, relu):
    return x.new_empty((x.shape[0], w.shape[0]))


# ---- PoseDistanceLoss ------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_NS + "::pose_loss", mutates_args=(), device_types="cuda")
def pose_loss(pred: torch.Tensor, truth: torch.Tensor, metric: int, mode: int, scale: float, alpha: float, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> ([loss, val position error, val orientation error] fp32, d loss / d pred); metric 0 l2, 1 l1, 2 linf, 3 combined; mode 0 position, 1 pose (the two validation metrics are always in elements 1, 2)"""
    out3, grad = _ops().pose_loss(pred, truth, metric, mode, scale, alpha, eps, want_grad=True)
    return out3, grad


@pose_loss.register_fake
def _(pred, truth, metric, mode, scale, alpha, eps):
    return pred.new_empty((3,)), torch.empty_like(pred)


@torch.library.custom_op(_NS + "::pose_distance_loss", mutates_args=(), device_types="cuda")
def pose_distance_loss(pred: torch.Tensor, truth: torch.Tensor, metric: int, mode: int, scale: float, alpha: float, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """differentiable in pred: -> (scalar loss, its gradient w.r.t. pred -- saved for the backward, not differentiable itself)"""
    out3, grad = _ops().pose_loss(pred, truth, metric, mode, scale, alpha, eps, want_grad=True)
    return out3[0].clone(), grad


@pose_distance_loss.register_fake
def _(pred, truth, metric, mode, scale, alpha, eps):
    return pred.new_empty(()), torch.empty_like(pred)


def _pdl_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _pdl_backward(ctx, dloss, _dgrad):
    (g,) = ctx.saved_tensors
    return g * dloss, None, None, None, None, None, None


pose_distance_loss.register_autograd(_pdl_backward, setup_context=_pdl_setup)


# ---- Adam ------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_NS + "::adam_step", mutates_args=("p", "m", "v"), device_types="cuda")
def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float, step: int) -> None:
    _ops().adam_step(p, g, m, v, lr, beta1, beta2, eps, step)
