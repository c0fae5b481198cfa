"""One-shot (non-temporal) pose regressors on the MI355X HIP path.

Drop-in for models/naive.py of the reference: same class names, constructor signatures,
attribute names, state_dict keys and return values; `compute_dtype` is the only addition
(keyword, last).  Forward/backward are explicit op compositions (see models/_core.py).
"""
import torch
import torch.nn as nn

from .. import ops
from ..headops import LinearOp, new_rows
from ._core import PoseModelBase, Replicated


class NaiveEndEffectorStateEstimator(PoseModelBase):
    """ResNet features -> pre-MLP (own eef pose) -> (pre - measurement) joined to the features ->
    post-MLP (other arm's eef pose).  reference: models/naive.py:8-127 (forward :68-112).
    Every Linear is followed by ReLU, the 7-d outputs included."""

    def __init__(self, hidden_dims_pre_measurement, hidden_dims_post_measurement, num_resnet_layers=50, latent_dim=50,
                 feature_extract=True, compute_dtype=None):
        super().__init__()
        # the reference calls import_resnet with its default use_pretrained=True here (naive.py:42)
        self._init_features(num_resnet_layers, latent_dim, feature_extract, True, None, False, wrap=False, register_heads=True,
                            compute_dtype=compute_dtype)
        pre = [latent_dim] + list(hidden_dims_pre_measurement) + [7]
        for i in range(len(pre) - 1):
            setattr(self, "pre_fc{}".format(i), nn.Linear(pre[i], pre[i + 1]))
        self.n_pre_hidden = len(pre) - 1
        post = [latent_dim + 7] + list(hidden_dims_post_measurement) + [7]
        for i in range(len(post) - 1):
            setattr(self, "post_fc{}".format(i), nn.Linear(post[i], post[i + 1]))
        self.n_post_hidden = len(post) - 1
        self._pre_ops = [LinearOp(getattr(self, "pre_fc%d" % i).weight, getattr(self, "pre_fc%d" % i).bias, relu=True)
                         for i in range(self.n_pre_hidden)]
        self._post_ops = [LinearOp(getattr(self, "post_fc%d" % i).weight, getattr(self, "post_fc%d" % i).bias, relu=True)
                          for i in range(self.n_post_hidden)]

    def forward(self, img, depth, self_measurement):
        """img (N,3,H,W), depth ignored, self_measurement (N,7) -> (pre_out (N,7), post_out (N,7))"""
        return self._call(img, depth, self_measurement)

    def _forward_impl(self, img, depth, x0bar, save):
        n, L = img.shape[0], self.latent_dim
        feat = new_rows(n, L, img.device)
        self._features_fwd(img, None, feat, save)
        h = feat
        for op in self._pre_ops:
            h = op.fwd(h, save=save)
        pre = h
        post_in = new_rows(n, L + 7, img.device)
        ops.copy2d(feat, post_in, cols=L)
        ops.copy2d((pre - x0bar).contiguous(), post_in[:, L:], cols=7)
        h = post_in
        for op in self._post_ops:
            h = op.fwd(h, save=save)
        return pre.contiguous(), h.contiguous()

    def _backward_impl(self, d_outs):
        L = self.latent_dim
        d_pre, d_post = d_outs
        dev = self._pre_ops[0].x.device
        n = self._pre_ops[0].x.shape[0]
        d = self._pad_rows(d_post, 7) if d_post is not None else new_rows(n, 7, dev)
        for op in reversed(self._post_ops):
            d = op.bwd(d)
        d_feat = new_rows(n, L, dev)
        ops.copy2d(d, d_feat, cols=L)
        d_pre_total = d[:, L:L + 7].contiguous()
        if d_pre is not None:
            d_pre_total = d_pre_total + d_pre.reshape(n, 7)
        d = self._pad_rows(d_pre_total, 7)
        for op in reversed(self._pre_ops):
            d = op.bwd(d)
        d_feat.add_(d)
        self._features_bwd(d_feat)

    def reset_initial_state(self, batch_size):
        """No temporal state (reference: models/naive.py:114-123)."""
        pass

    @property
    def requires_sequence(self):
        return False


class NaiveObjectStateEstimator(PoseModelBase):
    """ResNet features + early-feature aux head (+depth) + proprioception -> FC/ReLU stack -> object pose.
    reference: models/naive.py:130-367 (forward :298-352).  The final 7-d output is ReLU'd too."""

    def __init__(self, object_name, hidden_dims, num_resnet_layers=50, latent_dim=50, feature_extract=True,
                 feature_layer_nums=(9,), use_depth=False, use_pretrained=True, no_proprioception=False, compute_dtype=None):
        super().__init__()
        self.object_name = object_name
        self.use_proprioception = not no_proprioception
        self._init_features(num_resnet_layers, latent_dim, feature_extract, use_pretrained, feature_layer_nums, use_depth, wrap=True,
                            register_heads=True, compute_dtype=compute_dtype)
        print("Latent Dim + Aux Dim = {}".format(latent_dim + self.aux_latent_dim))
        if type(hidden_dims) is int:
            hidden_dims = [hidden_dims]
        input_dim = latent_dim + self.aux_latent_dim + (7 if self.use_proprioception else 0)
        fc_dims = [input_dim] + list(hidden_dims) + [7]
        for i in range(len(fc_dims) - 1):
            setattr(self, "fc{}".format(i), Replicated(nn.Linear(fc_dims[i], fc_dims[i + 1])))
        self.n_fc = len(fc_dims) - 1
        self.input_dim = input_dim
        self._fc_ops = [LinearOp(getattr(self, "fc%d" % i).module.weight, getattr(self, "fc%d" % i).module.bias, relu=True)
                        for i in range(self.n_fc)]

    def forward(self, img, depth, self_measurement):
        """img (N,3,H,W), depth (N,1,H,W) (read only when use_depth), self_measurement (N,7) -> (N,7)"""
        return self._call(img, depth, self_measurement)

    def _forward_impl(self, img, depth, x0bar, save):
        n = img.shape[0]
        rows = new_rows(n, self.input_dim, img.device)
        self._features_fwd(img, depth, rows, save)
        if self.use_proprioception:
            ops.copy2d(x0bar.reshape(n, 7), rows[:, self.latent_dim + self.aux_latent_dim:], cols=7)
        h = rows
        for op in self._fc_ops:
            h = op.fwd(h, save=save)
        return (h.contiguous(),)

    def _backward_impl(self, d_outs):
        d = self._pad_rows(d_outs[0], 7)
        for op in reversed(self._fc_ops):
            d = op.bwd(d)
        self._features_bwd(d)

    def reset_initial_state(self, batch_size):
        """No temporal state (reference: models/naive.py:354-363)."""
        pass

    @property
    def requires_sequence(self):
        return False
