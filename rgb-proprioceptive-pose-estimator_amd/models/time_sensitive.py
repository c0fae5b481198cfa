"""LSTM (temporally dependent) pose regressors on the MI355X HIP path.

Drop-in for models/time_sensitive.py of the reference: class names, constructor signatures
(`dropout_prob` and `device` accepted and unused, as there), attributes (`sequence_length`,
`rollout`, `requires_sequence`, `reset_initial_state`), state_dict keys and return values are kept.
Inputs are time-major (S, N, ...); all S*N frames go through the trunk as one batch
(time_sensitive.py:181-182) and the LSTMs start from zeros each chunk unless `rollout` is set,
in which case (h, c) is carried on the module between calls (time_sensitive.py:211-217,503-507).
"""
import torch
import torch.nn as nn

from .. import ops
from ..headops import LinearOp, LSTMOp, new_rows
from ._core import PoseModelBase, Replicated


def _lstm_op(lstm):
    return LSTMOp(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)


class _SequenceModel(PoseModelBase):
    def _seq_inputs(self, img, depth, x0bar):
        S, N = img.shape[0], img.shape[1]
        img = img.reshape(S * N, *img.shape[2:])
        if depth is not None and self.use_depth:
            depth = depth.reshape(S * N, *depth.shape[2:])
        else:
            depth = None
        return S, N, img, depth, None if x0bar is None else x0bar.reshape(S * N, 7)

    def _state(self, name, n, hid, device):
        """carried (h, c) in rollout mode, None (zero start) otherwise.  The state tensors are PERSISTENT (updated in place by
        _keep): a frame captured into a hipGraph (util.learn_utils.GraphedRolloutFrame) reads and writes fixed addresses."""
        if not self.rollout:
            return None, None
        st = self._carried.get(name)
        if st is None or st[0].shape != (n, hid) or st[0].device != device:
            st = (torch.zeros((n, hid), dtype=torch.float32, device=device), torch.zeros((n, hid), dtype=torch.float32, device=device))
            self._carried[name] = st
        return st

    def _keep(self, name, hc):
        if self.rollout:
            st = self._carried.get(name)
            if st is None or st[0].shape != hc[0].shape or st[0].device != hc[0].device:
                self._carried[name] = (hc[0].clone(), hc[1].clone())
            else:
                st[0].copy_(hc[0])
                st[1].copy_(hc[1])

    def _reset_carried(self):
        """zero initial state (reset_initial_state of the reference: time_sensitive.py:256-270,519-529,788-800), in place"""
        for h, c in self._carried.values():
            h.zero_()
            c.zero_()

    @property
    def requires_sequence(self):
        return True


class TemporallyDependentStateEstimator(_SequenceModel):
    """Two-arm model: features -> LSTM -> Linear = own eef pose; (pose - measurement) joined to the features
    -> LSTM -> Linear = other arm's pose.  reference: models/time_sensitive.py:8-274 (forward :165-254).
    Quirk kept: the aux/depth heads live in plain lists, so they are neither trained nor saved."""

    def __init__(self, hidden_dim_pre_measurement, hidden_dim_post_measurement, num_resnet_layers=50, latent_dim=50,
                 sequence_length=10, dropout_prob=0.10, feature_extract=True, feature_layer_nums=(9,), use_depth=False,
                 use_pretrained=True, device='cpu', compute_dtype=None):
        super().__init__()
        self._init_features(num_resnet_layers, latent_dim, feature_extract, use_pretrained, feature_layer_nums, use_depth, wrap=False,
                            register_heads=False, compute_dtype=compute_dtype)
        print("Latent Dim + Aux Dim = {}".format(latent_dim + self.aux_latent_dim))
        fdim = latent_dim + self.aux_latent_dim
        self.pre_measurement_rnn = nn.LSTM(input_size=fdim, hidden_size=hidden_dim_pre_measurement)
        self.pre_measurement_fc = nn.Linear(hidden_dim_pre_measurement, 7)
        self.post_measurement_rnn = nn.LSTM(input_size=fdim + 7, hidden_size=hidden_dim_post_measurement)
        self.post_measurement_fc = nn.Linear(hidden_dim_post_measurement, 7)
        self.sequence_length = sequence_length
        self.pre_measurement_hidden_dim = hidden_dim_pre_measurement
        self.post_measurement_hidden_dim = hidden_dim_post_measurement
        self.pre_out_vec = None
        self.post_out_vec = None
        self._carried = {}
        self._fdim = fdim
        self._pre_rnn, self._post_rnn = _lstm_op(self.pre_measurement_rnn), _lstm_op(self.post_measurement_rnn)
        self._pre_fc = LinearOp(self.pre_measurement_fc.weight, self.pre_measurement_fc.bias)
        self._post_fc = LinearOp(self.post_measurement_fc.weight, self.post_measurement_fc.bias)

    def forward(self, img, depth, self_measurement):
        """img (S,N,3,H,W), depth (S,N,1,H,W), self_measurement (S,N,7) -> (pre_out, post_out), each (S,N,7)"""
        return self._call(img, depth, self_measurement)

    def _forward_impl(self, img, depth, x0bar, save):
        S, N, img, depth, x0bar = self._seq_inputs(img, depth, x0bar)
        dev, F = img.device, self._fdim
        rows = new_rows(S * N, F, dev)
        self._features_fwd(img, depth, rows, save)
        h0, c0 = self._state("pre", N, self.pre_measurement_hidden_dim, dev)
        hp, hc = self._pre_rnn.fwd(rows, S, N, h0, c0, save=save)
        self._keep("pre", hc)
        pre = self._pre_fc.fwd(hp, save=save)
        post_in = new_rows(S * N, F + 7, dev)
        ops.copy2d(rows, post_in, cols=F)
        ops.copy2d((pre - x0bar).contiguous(), post_in[:, F:], cols=7)
        h0, c0 = self._state("post", N, self.post_measurement_hidden_dim, dev)
        hq, hc = self._post_rnn.fwd(post_in, S, N, h0, c0, save=save)
        self._keep("post", hc)
        post = self._post_fc.fwd(hq, save=save)
        self._sn = (S, N)
        return pre.contiguous().view(S, N, 7), post.contiguous().view(S, N, 7)

    def _backward_impl(self, d_outs):
        S, N = self._sn
        F = self._fdim
        d_pre, d_post = d_outs
        dev = self._pre_fc.x.device
        d = self._pad_rows(d_post, 7) if d_post is not None else new_rows(S * N, 7, dev)
        d = self._post_fc.bwd(d)
        d = self._post_rnn.bwd(d)  # [S*N, F+7]
        d_rows = new_rows(S * N, F, dev)
        ops.copy2d(d, d_rows, cols=F)
        d_pre_total = d[:, F:F + 7].contiguous()
        if d_pre is not None:
            d_pre_total = d_pre_total + d_pre.reshape(S * N, 7)
        d = self._pre_fc.bwd(self._pad_rows(d_pre_total, 7))
        d = self._pre_rnn.bwd(d)
        d_rows.add_(d)
        self._features_bwd(d_rows)

    def reset_initial_state(self, batch_size):
        """Zero the carried LSTM states (reference: models/time_sensitive.py:256-270)."""
        self._reset_carried()
        self.pre_out_vec = []
        self.post_out_vec = []


class TemporallyDependentObjectStateEstimator(_SequenceModel):
    """features (+aux, +proprioception) -> LSTM -> Linear(h, h/4) -> Linear(h/4, 7), no activation between.
    reference: models/time_sensitive.py:277-533 (forward :453-517)."""

    def __init__(self, object_name, hidden_dim, num_resnet_layers=50, latent_dim=50, sequence_length=10, dropout_prob=0.10,
                 feature_extract=True, feature_layer_nums=(9,), use_depth=False, use_pretrained=True, no_proprioception=False,
                 device='cpu', compute_dtype=None):
        super().__init__()
        self.object_name = object_name
        self.use_proprioception = not no_proprioception
        self._init_features(num_resnet_layers, latent_dim, feature_extract, use_pretrained, feature_layer_nums, use_depth, wrap=True,
                            register_heads=True, compute_dtype=compute_dtype)
        print("Latent Dim + Aux Dim = {}".format(latent_dim + self.aux_latent_dim))
        input_dim = latent_dim + self.aux_latent_dim + (7 if self.use_proprioception else 0)
        self.rnn = Replicated(nn.LSTM(input_size=input_dim, hidden_size=hidden_dim))
        self.fc = Replicated(nn.Sequential(nn.Linear(hidden_dim, int(hidden_dim // 4)), nn.Linear(int(hidden_dim // 4), 7)))
        self.sequence_length = sequence_length
        self.hidden_dim = hidden_dim
        self.input_dim = input_dim
        self.out_vec = None
        self._carried = {}
        self._rnn = _lstm_op(self.rnn.module)
        self._fc0 = LinearOp(self.fc.module[0].weight, self.fc.module[0].bias)
        self._fc1 = LinearOp(self.fc.module[1].weight, self.fc.module[1].bias)

    def forward(self, img, depth, self_measurement):
        """img (S,N,3,H,W), depth (S,N,1,H,W), self_measurement (S,N,7) -> (S,N,7)"""
        return self._call(img, depth, self_measurement)

    def _forward_impl(self, img, depth, x0bar, save):
        S, N, img, depth, x0bar = self._seq_inputs(img, depth, x0bar)
        dev = img.device
        rows = new_rows(S * N, self.input_dim, dev)
        self._features_fwd(img, depth, rows, save)
        if self.use_proprioception:
            ops.copy2d(x0bar, rows[:, self.latent_dim + self.aux_latent_dim:], cols=7)
        h0, c0 = self._state("rnn", N, self.hidden_dim, dev)
        h, hc = self._rnn.fwd(rows, S, N, h0, c0, save=save)
        self._keep("rnn", hc)
        out = self._fc1.fwd(self._fc0.fwd(h, save=save), save=save)
        self._sn = (S, N)
        return (out.contiguous().view(S, N, 7),)

    def _backward_impl(self, d_outs):
        d = self._fc1.bwd(self._pad_rows(d_outs[0], 7))
        d = self._fc0.bwd(d)
        d = self._rnn.bwd(d)
        self._features_bwd(d)

    def reset_initial_state(self, batch_size):
        """Zero the carried LSTM state (reference: models/time_sensitive.py:519-529)."""
        self._reset_carried()
        self.out_vec = []


class TemporallyDependentObjectStateEstimatorV2(_SequenceModel):
    """Separate LSTMs per modality (image features / proprioception), concatenated into a 2-layer FC.
    reference: models/time_sensitive.py:536-804 (forward :714-786)."""

    def __init__(self, object_name, img_hidden_dim, proprio_hidden_dim=64, num_resnet_layers=50, latent_dim=50, sequence_length=10,
                 dropout_prob=0.10, feature_extract=True, feature_layer_nums=(9,), use_depth=False, use_pretrained=True, device='cpu',
                 compute_dtype=None):
        super().__init__()
        self.object_name = object_name
        self._init_features(num_resnet_layers, latent_dim, feature_extract, use_pretrained, feature_layer_nums, use_depth, wrap=True,
                            register_heads=True, compute_dtype=compute_dtype)
        print("Latent Dim + Aux Dim = {}".format(latent_dim + self.aux_latent_dim))
        input_dim = latent_dim + self.aux_latent_dim
        self.img_rnn = Replicated(nn.LSTM(input_size=input_dim, hidden_size=img_hidden_dim))
        self.proprio_rnn = Replicated(nn.LSTM(input_size=7, hidden_size=proprio_hidden_dim))
        fc_in = img_hidden_dim + proprio_hidden_dim
        self.fc = Replicated(nn.Sequential(nn.Linear(fc_in, int(fc_in // 4)), nn.Linear(int(fc_in // 4), 7)))
        self.sequence_length = sequence_length
        self.img_hidden_dim = img_hidden_dim
        self.proprio_hidden_dim = proprio_hidden_dim
        self.input_dim = input_dim
        self.out_vec = None
        self._carried = {}
        self._img_rnn, self._prop_rnn = _lstm_op(self.img_rnn.module), _lstm_op(self.proprio_rnn.module)
        self._fc0 = LinearOp(self.fc.module[0].weight, self.fc.module[0].bias)
        self._fc1 = LinearOp(self.fc.module[1].weight, self.fc.module[1].bias)

    def forward(self, img, depth, self_measurement):
        """img (S,N,3,H,W), depth (S,N,1,H,W), self_measurement (S,N,7) -> (S,N,7)"""
        return self._call(img, depth, self_measurement)

    def _forward_impl(self, img, depth, x0bar, save):
        S, N, img, depth, x0bar = self._seq_inputs(img, depth, x0bar)
        dev = img.device
        Hi, Hp = self.img_hidden_dim, self.proprio_hidden_dim
        rows = new_rows(S * N, self.input_dim, dev)
        self._features_fwd(img, depth, rows, save)
        h0, c0 = self._state("img", N, Hi, dev)
        hi, hc = self._img_rnn.fwd(rows, S, N, h0, c0, save=save)
        self._keep("img", hc)
        prop = self._pad_rows(x0bar, 7)
        h0, c0 = self._state("proprio", N, Hp, dev)
        hp, hc = self._prop_rnn.fwd(prop, S, N, h0, c0, save=save)
        self._keep("proprio", hc)
        cat = new_rows(S * N, Hi + Hp, dev)
        ops.copy2d(hi, cat, cols=Hi)
        ops.copy2d(hp, cat[:, Hi:], cols=Hp)
        out = self._fc1.fwd(self._fc0.fwd(cat, save=save), save=save)
        self._sn = (S, N)
        return (out.contiguous().view(S, N, 7),)

    def _backward_impl(self, d_outs):
        Hi, Hp = self.img_hidden_dim, self.proprio_hidden_dim
        d = self._fc1.bwd(self._pad_rows(d_outs[0], 7))
        d = self._fc0.bwd(d)  # [S*N, Hi+Hp]
        d_hi = d[:, :Hi].contiguous()
        d_hp = d[:, Hi:Hi + Hp].contiguous()
        self._prop_rnn.bwd(d_hp, need_dx=False)
        self._features_bwd(self._img_rnn.bwd(d_hi))

    def reset_initial_state(self, batch_size):
        """Zero the carried LSTM states (reference: models/time_sensitive.py:788-800)."""
        self._reset_carried()
        self.out_vec = []
