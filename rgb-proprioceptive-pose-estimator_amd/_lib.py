"""ctypes binding of librpe_hip.so (the C ABI declared in include/rpe_hip.h).

The library is the product: if it is missing or does not export a declared symbol
the import of this module raises -- there is no eager / CPU fallback anywhere in the
package.
"""
import ctypes
import os

# torch first: its wheel bundles its own libamdhip64.so.7.  Loading librpe_hip.so before torch would pull in
# /opt/rocm's copy of the same SONAME and leave the process with two HIP runtimes (ours then sees no device).
import torch  # noqa: F401
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RPE_LIB_PATH: another build of the SAME ABI, for A/B timing of two library builds on one box -- tools/ab_lib.sh)
LIB_PATH = os.environ.get("RPE_LIB_PATH") or os.path.join(_HERE, "librpe_hip.so")

RPE_F32, RPE_BF16, RPE_F16 = 0, 1, 2
ABI_VERSION = 3


class RpeError(RuntimeError):
    """A librpe_hip.so entry point returned a non-zero status."""


class BnBwdEpilogue(Structure):
    _fields_ = [(n, c_void_p) for n in ("y", "a_out", "mean", "invstd", "scale", "shift", "stats_part", "a_mask")]


class ResizePlan(Structure):
    _fields_ = [(n, c_int) for n in ("Hr", "Wr", "top", "left", "ksx", "ksy")] + [(n, c_void_p) for n in ("xb", "xk", "yb", "yk", "tmp")]


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ("batch", "in_h", "in_w", "in_c", "out_c", "kh", "kw", "stride", "pad")]


P, I, L, F, D = c_void_p, c_int, c_long, c_float, c_double
PD = POINTER(ConvDesc)

# name -> (restype, argtypes).  Status-returning functions (restype int) are wrapped to raise.
_SPEC = {
    "rpe_abi_version": (I, []),
    "rpe_last_error": (c_char_p, []),
    "rpe_build_id": (c_char_p, []),
    "rpe_x4_bytes": (L, [I, I, I, I]),
    "rpe_conv_out_hw": (I, [PD, POINTER(c_int), POINTER(c_int)]),
    "rpe_conv_stats_tiles": (L, [L]),
    "rpe_conv2d_fwd": (I, [PD, I, P, P, P, P, P]),
    "rpe_conv2d_fwd_affine": (I, [PD, I, P, P, P, P, P, I, P]),
    "rpe_conv2d_fwd_affine_workspace_bytes": (L, [PD, I]),
    "rpe_conv2d_fwd_affine_ws": (I, [PD, I, P, P, P, P, P, I, P, L, P]),
    "rpe_conv2d_dgrad": (I, [PD, I, P, P, P, P, P]),
    "rpe_conv2d_dgrad_stats_tiles": (L, [PD, I]),
    "rpe_conv2d_fwd_stats_tiles": (L, [PD, I]),
    "rpe_conv2d_dgrad_bn": (I, [PD, I, P, P, P, P, POINTER(BnBwdEpilogue), P]),
    "rpe_bn_backward_reduce": (I, [I, P, P, P, P, P, P, P, P, L, I, P, L, P, P, P]),
    "rpe_bn_backward_from_dz": (I, [I, P, P, P, P, P, P, I, P, P, P, L, I, P, P, P]),
    "rpe_bn_bwd_fold_scratch_bytes": (L, [I, I, I]),
    "rpe_bn_bwd_fold_conv1x1": (I, [I, I, I, P, P, P, P, P, P, P, P, P, L, P]),
    "rpe_conv1x1_dgrad_kcat": (I, [PD, I, P, P, P, P, P, POINTER(BnBwdEpilogue), P]),
    "rpe_bn_bwd_fold_y_conv1x1": (I, [I, I, I, P, P, P, P, P, P, P, P]),
    "rpe_conv1x1_dgrad_kcat_y": (I, [PD, I, P, P, P, P, P, P, POINTER(BnBwdEpilogue), P]),
    "rpe_conv1x1_wgrad_folded_y_scratch_bytes": (L, [PD, I]),
    "rpe_conv1x1_wgrad_folded_y": (I, [PD, I, P, P, P, P, P, P, P, P, P, L, P]),
    "rpe_conv1x1_wgrad_folded_scratch_bytes": (L, [PD, I]),
    "rpe_conv1x1_wgrad_folded": (I, [PD, I, P, P, P, P, P, P, P, P, P, L, P]),
    "rpe_bn_backward_coeffs": (I, [P, I, I, L, P, P, P, P, P]),
    "rpe_bn_backward_apply_dz": (I, [I, P, P, P, P, P, P, P, L, I, P]),
    "rpe_conv2d_wgrad": (I, [PD, I, P, P, P, P]),
    "rpe_conv2d_wgrad_workspace_bytes": (L, [PD, I]),
    "rpe_conv2d_wgrad_det": (I, [PD, I, P, P, P, P, L, P]),
    "rpe_stem_conv_fwd": (I, [I, P, P, P, P, I, I, I, P]),
    "rpe_stem_conv_fwd_affine": (I, [I, P, P, P, P, I, I, I, I, P]),
    "rpe_stem_conv_wgrad": (I, [I, P, P, P, I, I, I, P]),
    "rpe_stem_conv_wgrad_workspace_bytes": (L, [I, I, I, I]),
    "rpe_stem_conv_wgrad_det": (I, [I, P, P, P, I, I, I, P, L, P]),
    "rpe_pack_conv_weight": (I, [I, P, P, P, I, I, I, I, P]),
    "rpe_pack_conv_weights_multi": (I, [I, P, I, L, P]),
    "rpe_pack_stem_weight": (I, [I, P, P, P, P]),
    "rpe_unpack_stem_grad": (I, [P, P, P]),
    "rpe_stage_image_nhwc4": (I, [I, P, P, I, I, I, P]),
    "rpe_stage_frames_u8": (I, [I, P, P, I, I, I, I, I, POINTER(c_float), POINTER(c_float), P]),
    "rpe_stage_frames_u8_resized": (I, [I, P, P, I, I, I, I, I, I, I, I, I, P, P, I, P, P, I, P, POINTER(c_float), POINTER(c_float), P]),
    "rpe_bn_finalize": (I, [P, I, I, L, P, P, P, P, P, F, F, P, P, P, P, P, P]),
    "rpe_bn_eval_affine": (I, [I, P, P, P, P, F, P, P, P]),
    "rpe_bn_apply": (I, [I, P, P, P, P, P, L, I, I, P]),
    "rpe_bn_apply_mask": (I, [I, P, P, P, P, P, L, I, P, P]),
    "rpe_conv1x1_dgrad_bn_t_workspace_bytes": (L, [PD, I]),
    "rpe_conv1x1_dgrad_bn_t": (I, [PD, I, P, P, P, P, POINTER(BnBwdEpilogue), P, I, P, P, L, P]),
    "rpe_bn_backward_coeffs_t": (I, [I, P, I, I, L, P, P, I, P, P, P, P, P, P, P]),
    "rpe_conv1x1_wgrad_combine": (I, [PD, P, P, P, P, P, P, P, P, P]),
    "rpe_bn_apply_gram_workspace_bytes": (L, [I, L, I]),
    "rpe_bn_apply_gram": (I, [I, P, P, P, P, L, I, P, P, L, P]),
    "rpe_gram_ones_row": (L, [I]),
    "rpe_gram_workspace_bytes": (L, [I, L, I]),
    "rpe_gram": (I, [I, P, L, I, P, P, L, P]),
    "rpe_bn_stats_from_gram": (I, [I, P, I, I, P, I, L, P, P, P, P, P, c_float, c_float, P, P, P, P, P]),
    "rpe_conv1x1_fwd_bn": (I, [PD, I, P, P, P, P, P, P, P, P, P, P, P]),
    "rpe_bn_apply_res_bn": (I, [I, P, P, P, P, P, P, P, L, I, I, P, P]),
    "rpe_bn_backward": (I, [I, P, P, P, P, P, P, P, P, P, P, L, I, P, L, P, P, P]),
    "rpe_maxpool3x3s2_fwd": (I, [I, P, P, P, I, I, I, I, P]),
    "rpe_bn_apply_maxpool3x3s2": (I, [I, P, P, P, P, P, P, I, I, I, I, P]),
    "rpe_maxpool3x3s2_bwd": (I, [I, P, P, P, P, I, I, I, I, P]),
    "rpe_stem_bwd": (I, [I, P, P, P, P, P, P, P, P, P, L, P, P, P, P, P, P, I, I, I, P, L, P, P, P]),
    "rpe_avgpool_fwd": (I, [I, P, P, I, I, I, P]),
    "rpe_avgpool_bwd": (I, [I, P, P, I, I, I, P]),
    "rpe_aux_head_fwd": (I, [I, P, P, P, P, P, L, P, P, I, I, I, P]),
    "rpe_aux_head_bwd": (I, [I, P, L, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "rpe_aux_head_bwd_workspace_floats": (L, [I, I, I, I]),
    "rpe_aux_head_bwd_det": (I, [I, P, L, P, P, P, P, P, P, P, P, P, I, I, I, P, L, P]),
    "rpe_depth_head_fwd": (I, [P, P, P, P, P, I, I, I, P]),
    "rpe_depth_head_bwd": (I, [P, P, L, P, P, P]),
    "rpe_linear_fwd": (I, [I, P, I, P, I, P, P, I, I, I, I, I, P, I, P]),
    "rpe_linear_fwd_workspace_bytes": (L, [I, I, I, I]),
    "rpe_linear_fwd_ws": (I, [I, P, I, P, I, P, P, I, I, I, I, I, P, I, P, L, P]),
    "rpe_linear_wgrad": (I, [I, P, I, P, I, P, I, I, I, I, P]),
    "rpe_linear_wgrad_workspace_bytes": (L, [I, I, I, I]),
    "rpe_linear_wgrad_det": (I, [I, P, I, P, I, P, I, I, I, I, I, P, L, P]),
    "rpe_transpose_f32": (I, [P, P, I, I, I, I, P]),
    "rpe_relu_bwd": (I, [P, P, P, L, P]),
    "rpe_colsum": (I, [P, L, I, I, P, I, P]),
    "rpe_copy2d": (I, [P, I, P, I, L, I, P]),
    "rpe_lstm_cell_fwd": (I, [P, P, P, P, P, P, I, I, P]),
    "rpe_lstm_cell_bwd": (I, [P, P, P, P, P, P, I, I, P]),
    "rpe_pose_loss": (I, [P, P, L, I, I, F, F, F, P, P, P]),
    "rpe_adam_step": (I, [P, P, P, P, L, D, D, D, D, I, P]),
    "rpe_amp_unscale": (I, [P, L, P, P]),
    "rpe_amp_update": (I, [P, F, F, I, P]),
    "rpe_adam_step_amp": (I, [P, P, P, P, L, D, D, D, D, P, P]),
    "rpe_resnet50_create": (I, [POINTER(c_void_p), I, I, I, I, I]),
    "rpe_resnet_create": (I, [POINTER(c_void_p), I, I, I, I, I, I]),
    "rpe_resnet50_destroy": (None, [P]),
    "rpe_resnet50_workspace_bytes": (L, [P]),
    "rpe_resnet50_param_name": (c_char_p, [P, I]),
    "rpe_resnet50_param_numel": (L, [P, I]),
    "rpe_resnet50_bind": (I, [P, P, L, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "rpe_resnet50_pack_weights": (I, [P, P]),
    "rpe_resnet50_weights_changed": (I, [P]),
    "rpe_resnet50_forward": (I, [P, P, P, L, I, P]),
    "rpe_resnet50_forward_u8": (I, [P, P, I, I, POINTER(c_float), POINTER(c_float), P, L, I, P]),
    "rpe_resnet50_forward_u8_resized": (I, [P, P, I, I, POINTER(ResizePlan), POINTER(c_float), POINTER(c_float), P, L, I, P]),
    "rpe_resnet50_early_feature": (c_void_p, [P]),
    "rpe_resnet50_early_grad": (c_void_p, [P]),
    "rpe_resnet50_backward": (I, [P, P, L, I, P]),
    "rpe_resnet50_profile": (I, [P, I]),
    "rpe_resnet50_profile_read": (I, [P, POINTER(c_float), POINTER(c_int), POINTER(c_double), POINTER(c_double)]),
    "rpe_resnet50_backward_begin": (I, [P, P, L, P]),
    "rpe_resnet50_backward_blocks": (I, [P, I, I, P]),
    "rpe_resnet50_backward_end": (I, [P, I, P]),
    "rpe_resnet50_backward_frozen": (I, [P, P, L, P]),
    "rpe_resnet50_set_aux_grad": (I, [P, P, L, P, P, P]),
    "rpe_resnet50_profile_kernels": (L, [P, P, L]),
    "rpe_last_kernel_name": (c_char_p, []),
    "rpe_set_walk_direction": (None, [I]),
    "rpe_conv2d_wgrad_halo_min_width": (I, [I]),
    "rpe_resnet50_set_aux_head": (I, [P, P, P, P, P, L, P, P]),
    "rpe_resnet50_aux_head_bwd": (I, [P, P, L, P, P, P, P, P, P, P, P, L, P]),
    "rpe_aux_head_bwd_det_y": (I, [I, P, L, P, P, P, P, P, P, P, P, P, P, I, I, I, P, L, P]),
    "rpe_bn_apply_maxpool3x3s2_aux": (I, [I, P, P, P, P, P, P, I, I, I, P, P, P, P, L, P, P, P]),
    "rpe_resnet50_side_stream_info": (I, [P, POINTER(c_int), POINTER(c_int)]),
    "rpe_resnet50_tensor": (I, [P, c_char_p, POINTER(c_void_p), POINTER(c_long), POINTER(c_int)]),
    "rpe_resnet50_set_hook_grad": (I, [P, I, P]),
    "rpe_resnet50_set_stem_raw": (I, [P, I]),
    "rpe_aux_head_fwd_c": (I, [I, P, I, P, P, P, P, L, P, P, I, I, I, P]),
    "rpe_aux_head_bwd_c": (I, [I, P, L, P, I, P, P, P, P, P, P, P, P, I, I, I, P]),
    "rpe_depth_head_fwd_pools": (I, [P, P, P, P, P, I, I, I, I, P]),
    "rpe_tensor_add": (I, [I, P, P, L, P]),
}
# entry points whose int return value is data, not a status
_NOT_STATUS = {"rpe_abi_version", "rpe_conv2d_wgrad_halo_min_width"}

EXPORTS = tuple(_SPEC)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librpe_hip.so not found at %s -- build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); there is no fallback path." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SPEC.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("librpe_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.rpe_abi_version() != ABI_VERSION:
        raise ImportError("librpe_hip.so ABI version %d != expected %d" % (lib.rpe_abi_version(), ABI_VERSION))
    return lib


_raw = _load()


class _Checked:
    """Attribute access returns a callable that raises RpeError on a non-zero status."""

    def __getattr__(self, name):
        fn = getattr(_raw, name)
        res = _SPEC[name][0]
        if res is I and name not in _NOT_STATUS:

            def call(*a, _fn=fn, _name=name):
                rc = _fn(*a)
                if rc != 0:
                    raise RpeError("%s failed (%d): %s" % (_name, rc, _raw.rpe_last_error().decode()))
                return 0

            setattr(self, name, call)
            return call
        setattr(self, name, fn)
        return fn


lib = _Checked()
raw = _raw
