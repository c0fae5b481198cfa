"""PoseDistanceLoss on the MI355X HIP path.

Drop-in for models/losses.py:11-128 of the reference: same constructor, same ValueErrors, same
return types (0-d tensor in the training modes; (numpy scalar, float) in "val" mode).  One kernel
launch computes the summed loss, its gradient w.r.t. the prediction and both validation metrics;
the "val" numpy loop of the reference (losses.py:95-113) becomes a device-side reduction, with
`forward_device` exposing it without the host synchronisation.
"""
import torch
import torch.nn as nn

from .. import ops

DISTANCE_METRICS = {"l1", "l2", "linf", "combined"}
POSE_LOSS_MODES = {"position", "pose", "val"}
_METRIC_CODE = {"l2": 0, "l1": 1, "linf": 2, "combined": 3}


class _PoseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, truth, metric, mode, scale, alpha, eps):
        out3, grad = ops.pose_loss(prediction, truth, metric, mode, scale, alpha, eps, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.shape = prediction.shape
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None, None, None, None, None


class PoseDistanceLoss(nn.Module):
    """Summed position distance + alpha * (quaternion distance + negative-w penalty).

    distance_metric: "l2" (sum_i sqrt(|dp_i|^2 + epsilon)), "l1", "linf" or "combined" (their sum)
    mode: "position" | "pose" | "val" (returns summed position error and summed |angle| error in radians)
    The result is a SUM over all leading dimensions (losses.py:75,80,118,122), not a mean.
    """

    def __init__(self, distance_metric="l2", scale_factor=1.0, alpha=1.0, epsilon=1e-4, mode="pose"):
        super(PoseDistanceLoss, self).__init__()
        if distance_metric not in DISTANCE_METRICS:
            raise ValueError("Invalid distance metric specified; available are: {}, requested {}.".format(DISTANCE_METRICS, distance_metric))
        if mode not in POSE_LOSS_MODES:
            raise ValueError("Invalid loss mode specified; available are: {}, requested {}.".format(POSE_LOSS_MODES, mode))
        self.distance_metric = distance_metric
        self.scale_factor = scale_factor
        self.alpha = alpha
        self.epsilon = epsilon
        self.mode = mode

    @staticmethod
    def _prep(prediction, truth):
        if not (prediction.is_cuda and truth.is_cuda):
            raise RuntimeError("PoseDistanceLoss runs on the MI355X HIP path only; there is no CPU fallback")
        if prediction.shape != truth.shape or prediction.shape[-1] != 7:
            raise ValueError("prediction and truth must both be (*, 7), got %s and %s" % (tuple(prediction.shape), tuple(truth.shape)))
        return prediction.contiguous().float(), truth.contiguous().float()

    def forward_device(self, prediction, truth):
        """(position error sum, |angle| error sum) as 0-d device tensors: the "val" quantities without the
        .cpu() synchronisation the reference forces every step (util/learn_utils.py:164,173)."""
        p, t = self._prep(prediction.detach(), truth)
        out3, _ = ops.pose_loss(p, t, _METRIC_CODE[self.distance_metric], 0, 1.0, 0.0, self.epsilon, want_grad=False)
        return out3[0], out3[2]

    def forward(self, prediction, truth):
        """prediction (*, 7) = (x,y,z,i,j,k,w) with an UNNORMALISED quaternion; truth (*, 7) with a unit, w >= 0 quaternion."""
        if self.mode == "val":
            pos, ang = self.forward_device(prediction, truth)
            return pos.cpu().numpy(), float(ang.item())
        p, t = self._prep(prediction, truth)
        return _PoseLossFn.apply(p, t, _METRIC_CODE[self.distance_metric], 1 if self.mode == "pose" else 0, float(self.scale_factor),
                                 float(self.alpha), float(self.epsilon))
