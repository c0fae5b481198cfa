"""GPU parity of every C-ABI kernel against a CPU fp32 torch statement of the same op (and the
oracle where it has one).  Tolerances: fp32 path 2e-5 of the reference's max magnitude (exact-fp32
MFMA, different summation order); bf16 path 2e-2 (8-bit mantissa inputs, fp32 accumulation); fp16 path 3e-3 (11 bits)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from rgb_proprioceptive_pose_estimator_amd import ops
    from oracle import pose_oracle as po

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def tol(dtype):
    return {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]


def rel_err(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), "non-finite values in the GPU result"
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def q(t, dtype):
    """round to the compute dtype and back (so the reference sees the same inputs)"""
    return t.to(dtype).float()


def nhwc(t):  # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ dense GEMM (MFMA layout check)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(200, 7, 36), (128, 128, 64), (300, 130, 200), (64, 70, 8), (1, 512, 2048), (5000, 130, 1032), (4100, 40, 1024)])
def test_linear_fwd_layout(dtype, m, n, k):
    g = torch.Generator().manual_seed(m * 1000 + n)
    kp = (k + 7) // 8 * 8
    x = torch.zeros(m, kp)
    w = torch.zeros(n, kp)
    x[:, :k] = torch.randn(m, k, generator=g)
    w[:, :k] = torch.randn(n, k, generator=g) + torch.arange(n).float()[:, None] * 0.01  # asymmetric
    b = torch.randn(n, generator=g)
    xd, wd = x.to(dtype).to(DEV), w.to(dtype).to(DEV)
    out = ops.linear_fwd(xd, wd, b.to(DEV), relu=True)
    ref = F.relu(q(x, dtype) @ q(w, dtype).t() + b)
    assert rel_err(out, ref) < tol(dtype)


@pytest.mark.parametrize("m,n,k", [(1, 1024, 3655), (1, 7, 64), (2, 130, 259), (3, 64, 1000), (5, 33, 4099), (8, 512, 2048), (7, 9, 1)])
def test_linear_fwd_few_rows(m, n, k):
    """Rollout frames: fp32 Linear with <= 8 rows runs on the per-column kernel (any row stride / alignment, bias, addend, ReLU)."""
    g = torch.Generator().manual_seed(m * 77 + n)
    xf, wf, af = torch.randn(m, k + 3, generator=g), torch.randn(n, k + 1, generator=g), torch.randn(m, n + 2, generator=g)
    x, w, add = xf[:, :k], wf[:, :k], af[:, :n]            # row strides that are not multiples of 4 elements
    b = torch.randn(n, generator=g)
    xd, wd, ad = xf.to(DEV)[:, :k], wf.to(DEV)[:, :k], af.to(DEV)[:, :n]
    assert xd.stride(0) == k + 3 and wd.stride(0) == k + 1
    ref = x.double() @ w.double().t() + b.double()
    out = ops.linear_fwd(xd, wd, b.to(DEV))
    assert rel_err(out, ref.float()) < 2e-6
    out = ops.linear_fwd(xd, wd, b.to(DEV), relu=True, addend=ad)
    assert rel_err(out, F.relu(ref + add.double()).float()) < 2e-6
    out = ops.linear_fwd(xd, wd, None)
    assert rel_err(out, (ref - b.double()).float()) < 2e-6
    # the same product through the MFMA tile kernel (9 rows: one more than the few-row limit), row by row
    x9 = torch.cat([x, x[:1].expand(9 - m, k)], 0) if m < 9 else x
    kp = (k + 7) // 8 * 8
    xp, wp = torch.zeros(9, kp), torch.zeros(n, kp)
    xp[:, :k], wp[:, :k] = x9, w
    tile = ops.linear_fwd(xp.to(DEV), wp.to(DEV), b.to(DEV))[:m]
    assert rel_err(ops.linear_fwd(xd, wd, b.to(DEV)), tile.cpu()) < 2e-6


@pytest.mark.parametrize("m,n,k", [(256, 1024, 3656), (256, 512, 2048), (128, 2048, 512), (64, 130, 1000), (300, 7, 4100)])
def test_linear_fwd_split_k(m, n, k):
    """fp32 Linear layers with few output tiles and a long reduction (the heads at a few hundred rows) split K over grid.y through a
    workspace (rpe_linear_fwd_ws): planned split taken, result = reference, repeats bitwise, any N."""
    from rgb_proprioceptive_pose_estimator_amd._lib import lib, RPE_F32
    g = torch.Generator().manual_seed(m + n + k)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    add = torch.randn(m, ops.pad4(n), generator=g)[:, :n]
    assert lib.rpe_linear_fwd_workspace_bytes(RPE_F32, m, n, k) > 0, "this shape is expected to split"
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ad = torch.zeros(m, ops.pad4(n)).to(DEV)
    ad[:, :n] = add.to(DEV)
    out = ops.linear_fwd(xd, wd, bd, relu=True, addend=ad[:, :n])
    assert "nt_split_epilogue_kernel" in ops.last_kernel_name()
    ref = F.relu(x.double() @ w.double().t() + b.double() + add.double()).float()
    assert rel_err(out, ref) < 2e-5
    assert torch.equal(out, ops.linear_fwd(xd, wd, bd, relu=True, addend=ad[:, :n]))
    plain = ops.linear_fwd(xd, wd, bd)
    assert rel_err(plain, (x.double() @ w.double().t() + b.double()).float()) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(333, 7, 36), (1000, 128, 128), (4096, 64, 200), (50, 130, 64)])
def test_linear_wgrad(dtype, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    npad, kp = (n + 7) // 8 * 8, (k + 7) // 8 * 8
    dy = torch.zeros(m, npad)
    x = torch.zeros(m, kp)
    dy[:, :n] = torch.randn(m, n, generator=g)
    x[:, :k] = torch.randn(m, k, generator=g) + torch.arange(k).float()[None, :] * 0.01
    dw = torch.zeros(n, kp, device=DEV)
    ops.linear_wgrad(dy.to(dtype).to(DEV), x.to(dtype).to(DEV), dw, n=n, k=kp)
    ref = q(dy, dtype)[:, :n].t() @ q(x, dtype)
    assert rel_err(dw, ref) < tol(dtype)
    ops.linear_wgrad(dy.to(dtype).to(DEV), x.to(dtype).to(DEV), dw, n=n, k=kp)        # accumulates: dw += ...
    assert rel_err(dw, 2 * ref) < tol(dtype)
    dwa = torch.zeros(n, kp, device=DEV)
    ops.linear_wgrad(dy.to(dtype).to(DEV), x.to(dtype).to(DEV), dwa, n=n, k=kp, deterministic=False)
    assert rel_err(dwa, ref) < tol(dtype)


# ------------------------------------------------------------------ convolution
CONVS = [  # (B, H, Cin, Cout, k, stride, pad)
    (2, 14, 64, 64, 3, 1, 1),
    (2, 14, 128, 64, 3, 2, 1),
    (3, 8, 64, 256, 1, 1, 0),
    (2, 14, 256, 512, 1, 2, 0),
    (1, 7, 512, 128, 3, 1, 1),
    (2, 10, 64, 128, 3, 2, 1),  # Ho*Wo not a multiple of anything convenient
    (2, 9, 64, 64, 3, 2, 1),    # odd input size: stride-2 data gradient without the parity-class decomposition
    (3, 12, 64, 128, 1, 2, 0),
    # M >= 1024 rows and K >= 1024: the 128-B-K-row / 2-slot-ring configuration (the 256-row / 8-wave tiles of round 1 were removed in round 3)
    (8, 24, 128, 128, 3, 1, 1),
    (8, 24, 1024, 64, 1, 1, 0),
    (6, 30, 128, 256, 3, 2, 1),
    # 3x3 / stride 1 / pad 1 in the 16-bit types run the halo form (tiles of whole output rows, activation patch resident in LDS):
    # several 64-channel chunks, tiles that straddle images, a last tile with fewer rows, a partial N tile, the widest rows that fit
    (3, 9, 128, 192, 3, 1, 1),
    (5, 7, 256, 64, 3, 1, 1),
    (37, 7, 64, 64, 3, 1, 1),
    (2, 56, 64, 64, 3, 1, 1),
    (4, 28, 128, 128, 3, 1, 1),
    (1, 3, 64, 128, 3, 1, 1),
    (2, 80, 64, 64, 3, 1, 1),
    (1, 100, 64, 64, 3, 1, 1),   # a row of 102 pixels x 3 does not fit the patch: gathered form
]


def _conv_inputs(cfg, dtype):
    b, h, ci, co, k, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(b, ci, h, h, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    return q(x, dtype), q(w, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_and_stats(dtype, cfg):
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    ref = F.conv2d(x, w, None, s, p)
    y, st = ops.conv2d_fwd(nhwc(x).to(dtype).to(DEV), nhwc(w).to(dtype).to(DEV), s, p, want_stats=True)
    assert rel_err(nchw(y), ref) < tol(dtype)
    sums = st.sum(0).cpu()  # [2, Co]
    assert rel_err(sums[0], ref.sum((0, 2, 3))) < 1e-3 + tol(dtype)
    assert rel_err(sums[1], (ref * ref).sum((0, 2, 3))) < 1e-3 + tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_dgrad(dtype, cfg):
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    g = torch.Generator().manual_seed(5)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    add = q(torch.randn(x.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, x, dy)
    w_crsk = w.permute(1, 2, 3, 0).contiguous()  # [Ci, kh, kw, Co]
    dx = ops.conv2d_dgrad(nhwc(dy).to(dtype).to(DEV), w_crsk.to(dtype).to(DEV), (b, h, h, ci), s, p, addend=nhwc(add).to(dtype).to(DEV))
    assert rel_err(nchw(dx), ref + add) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_wgrad(dtype, cfg):
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = q(torch.randn(y.shape, generator=torch.Generator().manual_seed(6)), dtype)
    (ref,) = torch.autograd.grad(y, w, dy)
    xd, dyd = nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV)
    dw = ops.conv2d_wgrad(xd, dyd, k, s, p)                      # slab + fixed-order sum
    assert rel_err(dw.permute(0, 3, 1, 2), ref) < tol(dtype)
    assert torch.equal(dw, ops.conv2d_wgrad(xd, dyd, k, s, p)), "the deterministic weight gradient is not bitwise reproducible"
    dwa = ops.conv2d_wgrad(xd, dyd, k, s, p, deterministic=False)   # fp32 atomics
    assert rel_err(dwa.permute(0, 3, 1, 2), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cfg", [(8, 28, 64, 64, 3, 1, 1), (6, 14, 256, 128, 1, 1, 0), (5, 30, 64, 128, 3, 2, 1)])
def test_walk_direction_does_not_change_results(dtype, cfg):
    """rpe_set_walk_direction is scheduling only: every XCD walks its share of a launch's row tiles / spans upwards (0), downwards (1) or
    alternately (2), and the conv forward (+ its BN partial sums), data gradient, weight gradient and the BatchNorm apply pass give the
    same bits either way (partial sums are indexed by tile, not by arrival)."""
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    xd, wd = nhwc(x).to(dtype).to(DEV), nhwc(w).to(dtype).to(DEV)
    ho = (h + 2 * p - k) // s + 1
    g = torch.Generator().manual_seed(9)
    dyd = nhwc(q(torch.randn(b, co, ho, ho, generator=g), dtype)).to(dtype).to(DEV)
    w_crsk = w.permute(1, 2, 3, 0).contiguous().to(dtype).to(DEV)
    sc, sh = (torch.rand(co, generator=g) + 0.5).to(DEV), (torch.randn(co, generator=g) * 0.3).to(DEV)

    def run():
        y, st = ops.conv2d_fwd(xd, wd, s, p, want_stats=True)
        return (y, st, ops.conv2d_dgrad(dyd, w_crsk, (b, h, h, ci), s, p), ops.conv2d_wgrad(xd, dyd, k, s, p), ops.bn_apply(y, sc, sh),
                ops.bn_apply(y, sc, sh), ops.conv2d_fwd(xd, wd, s, p, want_stats=True)[0])   # (mode 2: consecutive launches differ in direction)

    ref = run()
    try:
        for mode in (1, 2):
            ops.set_walk_direction(mode)
            got = run()
            for a, r in zip(got, ref):
                assert torch.equal(a, r), "walk direction %d changed a result" % mode
    finally:
        ops.set_walk_direction(0)


def test_conv_wgrad_many_splits_is_deterministic():
    """A long reduction (many M splits per tile, the shape class of the layer1 1x1 weight gradients at bs256): slab mode must
    reproduce itself bitwise and agree with the atomic mode to fp32 summation-order noise."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 56, 56, 64, generator=g).bfloat16().to(DEV)
    dy = torch.randn(8, 56, 56, 256, generator=g).bfloat16().to(DEV)
    a = ops.conv2d_wgrad(x, dy, 1, 1, 0)
    b = ops.conv2d_wgrad(x, dy, 1, 1, 0)
    c = ops.conv2d_wgrad(x, dy, 1, 1, 0, deterministic=False)
    assert torch.equal(a, b)
    assert rel_err(a, c) < 1e-5
    ref = dy.float().reshape(-1, 256).t() @ x.float().reshape(-1, 64)
    assert rel_err(a.reshape(256, 64), ref) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mask_mode", [1, 2, 4])
@pytest.mark.parametrize("cfg", [(2, 14, 64, 64, 3, 1, 1), (2, 14, 256, 512, 1, 2, 0), (3, 8, 128, 256, 1, 1, 0), (8, 24, 128, 128, 3, 1, 1),
                                 (6, 30, 64, 1024, 1, 1, 0)])
def test_dgrad_with_fused_bn_backward(dtype, mask_mode, cfg):
    """conv data-gradient + ReLU mask + BN-backward reduction in one epilogue == torch's unfused composition."""
    b, h, ci, co, k, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg) + mask_mode)
    # the producing layer: y_prev -> BN (+res) -> ReLU = a_prev, which is this conv's input
    y_prev = q(torch.randn(b, ci, h, h, generator=g) * 1.5 + 0.3, dtype).requires_grad_(True)
    if mask_mode == 4 and dtype == torch.float32:
        pytest.skip("the packed ReLU mask exists for 16-bit activations only")
    res = q(torch.randn(b, ci, h, h, generator=g), dtype) if mask_mode in (1, 4) else None
    gamma = (0.5 + torch.rand(ci, generator=g)).requires_grad_(True)
    beta = (torch.rand(ci, generator=g) - 0.5).requires_grad_(True)
    z = F.batch_norm(y_prev, None, None, gamma, beta, True, 0.1, 1e-5)
    a_prev = F.relu(z + res) if res is not None else F.relu(z)
    w = q(torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5, dtype)
    yc = F.conv2d(a_prev, w, None, s, p)
    dyc = q(torch.randn(yc.shape, generator=g), dtype)
    add = q(torch.randn(a_prev.shape, generator=g), dtype)
    dy_ref, dg_ref, db_ref = torch.autograd.grad([yc, a_prev], (y_prev, gamma, beta), [dyc, add])
    # GPU
    yn = nhwc(y_prev.detach())
    rows = yn.numel() // ci
    mean = yn.reshape(rows, ci).mean(0)
    var = yn.reshape(rows, ci).var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.detach() * invstd
    shift = beta.detach() - mean * scale
    yd = yn.to(dtype).to(DEV)
    resd = None if res is None else nhwc(res).to(dtype).to(DEV)
    a_mask = None
    if mask_mode == 4:   # packed ReLU mask written by the forward's bn_apply instead of re-reading a_out
        a_d, a_mask = ops.bn_apply_mask(yd, scale.to(DEV), shift.to(DEV), resd)
        assert torch.equal(a_d, ops.bn_apply(yd, scale.to(DEV), shift.to(DEV), resd, relu=True))
    else:
        a_d = ops.bn_apply(yd, scale.to(DEV), shift.to(DEV), resd, relu=True)
    w_crsk = w.permute(1, 2, 3, 0).contiguous().to(dtype).to(DEV)
    dz, st = ops.conv2d_dgrad_bn(nhwc(dyc).to(dtype).to(DEV), w_crsk, (b, h, h, ci), s, p, yd, mean.to(DEV), invstd.to(DEV),
                                 a_out=a_d if mask_mode == 1 else None, scale=scale.to(DEV) if mask_mode == 2 else None,
                                 shift=shift.to(DEV) if mask_mode == 2 else None, addend=nhwc(add).to(dtype).to(DEV), a_mask=a_mask)
    dy, dg, db = ops.bn_backward_from_dz(dz, yd, mean.to(DEV), invstd.to(DEV), gamma.detach().to(DEV), st)
    t = {torch.float32: 1e-4, torch.bfloat16: 3e-2, torch.float16: 5e-3}[dtype]
    assert rel_err(nchw(dy), dy_ref) < t
    assert rel_err(dg, dg_ref) < t and rel_err(db, db_ref) < t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 14, 64, 256), (3, 10, 128, 512), (5, 24, 64, 256), (2, 7, 512, 2048), (4, 14, 256, 1024)])
@pytest.mark.parametrize("epilogue", [False, True])
def test_bn_backward_folded_into_conv1x1_dgrad(dtype, cfg, epilogue):
    """dx = dz (A o W) + a_in G + b (rpe_bn_bwd_fold_conv1x1 + rpe_conv1x1_dgrad_kcat) == torch's BatchNorm backward followed by
    the 1x1 conv data gradient; optionally with the fused ReLU-mask / BN-partial-sum epilogue of the layer behind."""
    b, h, ci, co = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    y_prev = q(torch.randn(b, ci, h, h, generator=g) * 1.3 + 0.2, dtype).requires_grad_(True)      # the layer behind: BN2 -> ReLU = a_in
    g2 = (0.5 + torch.rand(ci, generator=g)).requires_grad_(True)
    b2 = (torch.rand(ci, generator=g) - 0.5).requires_grad_(True)
    a_in_t = F.relu(F.batch_norm(y_prev, None, None, g2, b2, True, 0.1, 1e-5))
    a_in = q(a_in_t.detach(), dtype)
    a_leaf = a_in.clone().requires_grad_(True)
    w = q(torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5, dtype)
    y = q(F.conv2d(a_leaf, w), dtype)            # stored in the compute dtype, as the engine does
    y_leaf = y.detach().clone().requires_grad_(True)
    gamma = (0.5 + torch.rand(co, generator=g))
    beta = torch.rand(co, generator=g) - 0.5
    z = F.batch_norm(y_leaf, None, None, gamma, beta, True, 0.1, 1e-5)
    dz = q(torch.randn(z.shape, generator=g) + 0.3, dtype)
    (dy_ref,) = torch.autograd.grad(z, y_leaf, dz)
    dx_ref = F.conv_transpose2d(dy_ref, w)       # gradient of the 1x1 conv wrt a_in, from the UNROUNDED dy
    # GPU
    yn = nhwc(y.detach())
    rows = yn.numel() // co
    mean = yn.reshape(rows, co).mean(0)
    var = yn.reshape(rows, co).var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    dzn = nhwc(dz)
    xhat = (yn - mean) * invstd
    st = torch.stack([dzn.reshape(rows, co).sum(0), (dzn * xhat).reshape(rows, co).sum(0)])[None].contiguous()   # one "tile" of partial sums
    dgamma, dbeta, c1c2 = ops.bn_backward_coeffs(st.to(DEV), rows)
    assert rel_err(dbeta, dzn.reshape(rows, co).sum(0)) < 1e-5
    wf = w.reshape(co, ci).to(dtype).to(DEV).contiguous()
    wd = wf.t().contiguous()
    wk, bias = ops.bn_bwd_fold_conv1x1(wf, wd, gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2)
    a_d, dz_d = nhwc(a_in).to(dtype).to(DEV), dzn.to(dtype).to(DEV)
    t = {torch.float32: 1e-4, torch.bfloat16: 3e-2, torch.float16: 5e-3}[dtype]
    if not epilogue:
        dx = ops.conv1x1_dgrad_kcat(dz_d, a_d, wk, bias)
        assert rel_err(nchw(dx), dx_ref) < t
        # the streaming form of the same BN backward
        dy = ops.bn_backward_apply_dz(dz_d, yn.to(dtype).to(DEV), mean.to(DEV), invstd.to(DEV), gamma.to(DEV), c1c2)
        assert rel_err(nchw(dy), dy_ref) < t
        # and the folded weight gradient: dW = sum_m dy[m] a_in[m]^T from dz and a_in alone
        dw_ref = torch.einsum("bchw,bnhw->cn", dy_ref, a_in)
        dw = ops.conv1x1_wgrad_folded(dz_d, a_d, w.reshape(co, ci).contiguous().to(DEV), gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2)
        assert rel_err(dw, dw_ref) < (1e-3 if dtype == torch.float32 else t)
        assert torch.equal(dw, ops.conv1x1_wgrad_folded(dz_d, a_d, w.reshape(co, ci).contiguous().to(DEV), gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2))
        return
    # with the epilogue of the layer behind (mask recomputed from y_prev, BN2 partial sums)
    ypn = nhwc(y_prev.detach())
    m2 = ypn.reshape(rows, ci).mean(0)
    r2 = 1.0 / torch.sqrt(ypn.reshape(rows, ci).var(0, unbiased=False) + 1e-5)
    sc2 = g2.detach() * r2
    sh2 = b2.detach() - m2 * sc2
    dzin, st2 = ops.conv1x1_dgrad_kcat(dz_d, a_d, wk, bias, bn=dict(y=ypn.to(dtype).to(DEV), mean=m2.to(DEV), invstd=r2.to(DEV), scale=sc2.to(DEV), shift=sh2.to(DEV)))
    dyp, dg2, db2 = ops.bn_backward_from_dz(dzin, ypn.to(dtype).to(DEV), m2.to(DEV), r2.to(DEV), g2.detach().to(DEV), st2)
    dyp_ref, dg2_ref, db2_ref = torch.autograd.grad(a_in_t, (y_prev, g2, b2), dx_ref)
    assert rel_err(nchw(dyp), dyp_ref) < t
    assert rel_err(dg2, dg2_ref) < t and rel_err(db2, db2_ref) < t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 14, 256, 64), (3, 10, 512, 128), (5, 24, 256, 64), (2, 12, 64, 64), (2, 7, 1024, 256)])
@pytest.mark.parametrize("epilogue", [False, True])
def test_bn_backward_folded_into_the_conv_in_front_y_form(dtype, cfg, epilogue):
    """A Bottleneck's conv1 (x [Ci = 4p] -> y [Co = p]) with bn1's backward folded in (rpe_bn_bwd_fold_y_conv1x1 + rpe_conv1x1_dgrad_kcat_y +
    rpe_conv1x1_wgrad_folded_y): dx = [dz | y] [A o W ; C' o W] + b (+ shortcut gradient) and dW from dz, y, x alone == torch's BatchNorm
    backward followed by the conv's data and weight gradients; optionally with the fused epilogue of the layer behind (the previous
    block's bn3: a_out mask / packed mask, partial sums)."""
    b, h, ci, co = cfg
    g = torch.Generator().manual_seed(sum(cfg) + 1)
    rows = b * h * h
    x = q(torch.randn(b, ci, h, h, generator=g) * 0.8 + 0.3, dtype)
    w = q(torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5, dtype)
    y = q(F.conv2d(x, w), dtype)
    y_leaf = y.clone().requires_grad_(True)
    gamma, beta = 0.5 + torch.rand(co, generator=g), torch.rand(co, generator=g) - 0.5
    z = F.batch_norm(y_leaf, None, None, gamma, beta, True, 0.1, 1e-5)
    dz = q(torch.randn(z.shape, generator=g) + 0.2, dtype)
    (dy_ref,) = torch.autograd.grad(z, y_leaf, dz)
    add = q(torch.randn(b, ci, h, h, generator=g), dtype)
    dx_ref = F.conv_transpose2d(dy_ref, w) + add
    dw_ref = torch.einsum("bkhw,bnhw->kn", dy_ref, x)
    yn, dzn, xn = nhwc(y), nhwc(dz), nhwc(x)
    mean = yn.reshape(rows, co).mean(0)
    invstd = 1.0 / torch.sqrt(yn.reshape(rows, co).var(0, unbiased=False) + 1e-5)
    xhat = (yn - mean) * invstd
    st = torch.stack([dzn.reshape(rows, co).sum(0), (dzn * xhat).reshape(rows, co).sum(0)])[None].contiguous()
    _, _, c1c2 = ops.bn_backward_coeffs(st.to(DEV), rows)
    wd = w.reshape(co, ci).t().contiguous().to(dtype).to(DEV)
    wk, bias = ops.bn_bwd_fold_y_conv1x1(wd, gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2)
    dz_d, y_d, x_d, add_d = dzn.to(dtype).to(DEV), yn.to(dtype).to(DEV), xn.to(dtype).to(DEV), nhwc(add).to(dtype).to(DEV)
    t = {torch.float32: 1e-4, torch.bfloat16: 3e-2, torch.float16: 5e-3}[dtype]
    if not epilogue:
        dx = ops.conv1x1_dgrad_kcat_y(dz_d, y_d, wk, bias, ci, addend=add_d)
        assert rel_err(nchw(dx), dx_ref) < t
        dx0 = ops.conv1x1_dgrad_kcat_y(dz_d, y_d, wk, bias, ci)               # without the shortcut gradient
        assert rel_err(nchw(dx0), dx_ref - add) < t
        dw = ops.conv1x1_wgrad_folded_y(dz_d, y_d, x_d, gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2)
        assert rel_err(dw, dw_ref) < (1e-3 if dtype == torch.float32 else t)
        assert torch.equal(dw, ops.conv1x1_wgrad_folded_y(dz_d, y_d, x_d, gamma.to(DEV), invstd.to(DEV), mean.to(DEV), c1c2))   # fixed-order sums
        return
    # the layer behind: x = relu(bn3(y3) + identity) of the previous block; dx above is the gradient wrt x
    y3 = q(torch.randn(b, ci, h, h, generator=g) * 1.2, dtype).requires_grad_(True)
    g3 = (0.5 + torch.rand(ci, generator=g)).requires_grad_(True)
    b3 = (torch.rand(ci, generator=g) - 0.5).requires_grad_(True)
    idn = q(torch.randn(b, ci, h, h, generator=g), dtype)
    out_prev = F.relu(F.batch_norm(y3, None, None, g3, b3, True, 0.1, 1e-5) + idn)
    dy3_ref, dg3_ref, db3_ref = torch.autograd.grad(out_prev, (y3, g3, b3), dx_ref)
    y3n = nhwc(y3.detach())
    m3 = y3n.reshape(rows, ci).mean(0)
    r3 = 1.0 / torch.sqrt(y3n.reshape(rows, ci).var(0, unbiased=False) + 1e-5)
    sc3, sh3 = g3.detach() * r3, b3.detach() - m3 * g3.detach() * r3
    y3d = y3n.to(dtype).to(DEV)
    bn = dict(y=y3d, mean=m3.to(DEV), invstd=r3.to(DEV))
    if dtype == torch.float32:
        bn["a_out"] = ops.bn_apply(y3d, sc3.to(DEV), sh3.to(DEV), nhwc(idn).to(dtype).to(DEV), relu=True)
    else:
        _, bn["a_mask"] = ops.bn_apply_mask(y3d, sc3.to(DEV), sh3.to(DEV), nhwc(idn).to(dtype).to(DEV))
    dz3, st3 = ops.conv1x1_dgrad_kcat_y(dz_d, y_d, wk, bias, ci, addend=add_d, bn=bn)
    dy3, dg3, db3 = ops.bn_backward_from_dz(dz3, y3d, m3.to(DEV), r3.to(DEV), g3.detach().to(DEV), st3)
    assert rel_err(nchw(dy3), dy3_ref) < t
    assert rel_err(dg3, dg3_ref) < t and rel_err(db3, db3_ref) < t


def test_pack_conv_weight():
    w = torch.randn(64, 3, 3, 128)
    wf, wd = ops.pack_conv_weight(w.to(DEV), torch.bfloat16)
    assert torch.equal(wf.cpu(), w.bfloat16())
    assert torch.equal(wd.cpu(), w.permute(3, 1, 2, 0).contiguous().bfloat16())


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_conv(dtype):
    g = torch.Generator().manual_seed(3)
    b, h = 2, 64
    img = q(torch.randn(b, 3, h, h, generator=g), dtype)
    w = q(torch.randn(64, 3, 7, 7, generator=g) / 12.0, dtype).requires_grad_(True)
    ref = F.conv2d(img, w, None, 2, 3)
    x4 = ops.stage_image(img.to(DEV), dtype)          # zero-bordered: [B, H + 6, W + 6, 4], the image at [3, 3 + H) x [3, 3 + W)
    assert x4.shape == (b, h + 6, h + 6, 4)
    inner = x4[:, 3:3 + h, 3:3 + h]
    assert torch.equal(inner[..., :3].float().cpu(), nhwc(img)) and (x4[..., 3] == 0).all()
    border = x4.clone()
    border[:, 3:3 + h, 3:3 + h] = 0
    assert not border.float().abs().max().item() > 0.0                         # nothing but zeros around it
    wp = ops.pack_stem_weight(w.detach().to(DEV), dtype)
    y, st = ops.stem_conv_fwd(x4, wp, want_stats=True)
    assert rel_err(nchw(y), ref) < tol(dtype)
    assert rel_err(st.sum(0)[0], ref.sum((0, 2, 3))) < 1e-3 + tol(dtype)
    dy = q(torch.randn(ref.shape, generator=g), dtype)
    (dw_ref,) = torch.autograd.grad(ref, w, dy)
    dw = ops.stem_conv_wgrad(x4, nhwc(dy).to(dtype).to(DEV))
    assert rel_err(dw, dw_ref) < tol(dtype)


# ------------------------------------------------------------------ batch norm / pooling
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,rows_hw", [(64, 14), (256, 7), (2048, 3)])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_train_fwd_bwd(dtype, c, rows_hw, residual):
    g = torch.Generator().manual_seed(c + rows_hw)
    b = 3
    y = q(torch.randn(b, c, rows_hw, rows_hw, generator=g) * 2 + 0.5, dtype).requires_grad_(True)
    res = q(torch.randn(b, c, rows_hw, rows_hw, generator=g), dtype) if residual else None
    gamma = (0.5 + torch.rand(c, generator=g)).requires_grad_(True)
    beta = (torch.rand(c, generator=g) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    z = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    a_ref = F.relu(z + res) if residual else F.relu(z)
    dA = q(torch.randn(a_ref.shape, generator=g), dtype)
    dy_ref, dg_ref, db_ref = torch.autograd.grad(a_ref, (y, gamma, beta), dA)
    # GPU: statistics from per-tile partials as the conv epilogue would produce them
    yn = nhwc(y.detach())
    rows = yn.numel() // c
    tiles = ops.stats_tiles(rows)
    flat = torch.zeros(tiles * 128, c)
    flat[:rows] = yn.reshape(rows, c)
    part = torch.stack([flat.view(tiles, 128, c).sum(1), (flat * flat).view(tiles, 128, c).sum(1)], 1).contiguous()
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    scale, shift, mean, invstd = ops.bn_finalize(part.to(DEV), rows, gamma.detach().to(DEV), beta.detach().to(DEV), rmd, rvd, nbt)
    assert rel_err(rmd, rm) < 1e-5 and rel_err(rvd, rv) < 1e-5 and nbt.item() == 1
    yd = yn.to(dtype).to(DEV)
    a = ops.bn_apply(yd, scale, shift, None if res is None else nhwc(res).to(dtype).to(DEV), relu=True)
    assert rel_err(nchw(a), a_ref) < tol(dtype)
    # backward uses the GPU's own (rounded) activation as the ReLU mask, like the engine does
    dy, dg, db, dz = ops.bn_backward(nhwc(dA).to(dtype).to(DEV), a, yd, mean, invstd, gamma.detach().to(DEV), want_dz=True)
    t = 5e-5 if dtype == torch.float32 else 3e-2
    assert rel_err(nchw(dy), dy_ref) < t
    assert rel_err(dg, dg_ref) < t and rel_err(db, db_ref) < t
    assert rel_err(nchw(dz), dA * (a_ref > 0)) < t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [192, 48, 2304])
def test_bn_streaming_kernels_any_channel_count(dtype, c):
    """The unrolled streaming BN kernels keep their per-channel coefficients in registers, which needs C / (16-byte chunk) to be a
    power of two; any other channel count (C = 192, ...) must take the form without unroll instead of being refused: rpe_bn_apply,
    rpe_bn_apply_mask, rpe_bn_apply_res_bn, rpe_bn_backward_apply_dz."""
    g = torch.Generator().manual_seed(c)
    rows = 700
    y, r = q(torch.randn(rows, c, generator=g) * 1.5 + 0.2, dtype), q(torch.randn(rows, c, generator=g), dtype)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rsc, rsh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    yd, rd = y.to(dtype).to(DEV), r.to(dtype).to(DEV)
    assert rel_err(ops.bn_apply(yd, sc.to(DEV), sh.to(DEV), rd, relu=True), F.relu(y * sc + sh + r)) < tol(dtype)
    assert rel_err(ops.bn_apply_res_bn(yd, sc.to(DEV), sh.to(DEV), rd, rsc.to(DEV), rsh.to(DEV)), F.relu(y * sc + sh + r * rsc + rsh)) < tol(dtype)
    if dtype != torch.float32:
        out, mask = ops.bn_apply_mask(yd, sc.to(DEV), sh.to(DEV), rd)
        bits = ((mask.cpu()[:, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(rows, c).bool()
        assert torch.equal(bits, out.cpu().float() > 0)
    mean, invstd, gamma = torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) + 0.5
    c1c2 = torch.randn(2, c, generator=g) * 0.1
    dz = q(torch.randn(rows, c, generator=g), dtype)
    ref = gamma * invstd * (dz - c1c2[0] - (y - mean) * invstd * c1c2[1])
    dy = ops.bn_backward_apply_dz(dz.to(dtype).to(DEV), yd, mean.to(DEV), invstd.to(DEV), gamma.to(DEV), c1c2.to(DEV))
    assert rel_err(dy, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_avgpool(dtype):
    g = torch.Generator().manual_seed(1)
    x = q(torch.randn(2, 64, 16, 16, generator=g), dtype).clamp_min(0).requires_grad_(True)  # post-ReLU-like (ties at 0)
    ref = F.max_pool2d(x, 3, 2, 1)
    out, idx = ops.maxpool_fwd(nhwc(x.detach()).to(dtype).to(DEV))
    assert rel_err(nchw(out), ref) == 0.0
    d = q(torch.randn(ref.shape, generator=g), dtype)
    add = q(torch.randn(x.shape, generator=g), dtype)
    (dref,) = torch.autograd.grad(ref, x, d)
    dx = ops.maxpool_bwd(nhwc(d).to(dtype).to(DEV), idx, (2, 16, 16, 64), addend=nhwc(add).to(dtype).to(DEV))
    # where x == 0 the winner among tied zeros is implementation-defined and irrelevant (ReLU kills it)
    mask = (x.detach() > 0).float()
    assert rel_err(nchw(dx) .cpu().float() * mask, (dref + add) * mask) < tol(dtype)
    x2 = q(torch.randn(3, 2048, 7, 7, generator=g), dtype)
    p = ops.avgpool_fwd(nhwc(x2).to(dtype).to(DEV))
    assert rel_err(p, x2.mean((2, 3))) < 1e-5
    for bb, hw in ((1, 7), (40, 7), (2, 3)):    # one frame (8 lanes per channel chunk) / many images / tiny maps (one lane per chunk)
        x3 = q(torch.randn(bb, 2048, hw, hw, generator=g), dtype)
        assert rel_err(ops.avgpool_fwd(nhwc(x3).to(dtype).to(DEV)), x3.mean((2, 3))) < 1e-5
    dp = torch.randn(3, 2048, generator=g)
    dx2 = ops.avgpool_bwd(dp.to(DEV), (3, 7, 7, 2048), dtype)
    assert rel_err(nchw(dx2), (dp / 49)[:, :, None, None].expand(3, 2048, 7, 7)) < tol(dtype)


# ------------------------------------------------------------------ heads
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,w,c", [(2, 16, 16, 64), (3, 6, 10, 64), (1, 2, 2, 8), (2, 112, 112, 64), (1, 8, 4, 128)])
def test_bn_apply_maxpool_fused_is_bitwise_the_two_passes(dtype, b, h, w, c):
    """the stem's BatchNorm apply + ReLU + max pool in one pass == rpe_bn_apply followed by rpe_maxpool3x3s2_fwd, bit for bit
    (activated map, pooled map and winner taps; ties and all-zero windows included: ReLU makes many)"""
    g = torch.Generator().manual_seed(b * 100 + h)
    y = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    scale = (0.5 + torch.rand(c, generator=g)).to(DEV)
    shift = (torch.rand(c, generator=g) - 0.7).to(DEV)
    a_ref = ops.bn_apply(y, scale, shift, None, relu=True)
    p_ref, i_ref = ops.maxpool_fwd(a_ref)
    a, p, i = ops.bn_apply_maxpool(y, scale, shift)
    assert torch.equal(a, a_ref) and torch.equal(p, p_ref) and torch.equal(i, i_ref)
    ref = F.max_pool2d(F.relu(nchw(y.float().cpu()) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]).to(dtype).float(), 3, 2, 1)
    assert rel_err(nchw(p), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_depth", [False, True])
def test_aux_and_depth_heads(dtype, use_depth):
    import ctypes
    from rgb_proprioceptive_pose_estimator_amd._lib import lib
    g = torch.Generator().manual_seed(9)
    b, h = 2, 16
    a1 = q(torch.randn(b, 64, h, h, generator=g), dtype).clamp_min(0).requires_grad_(True)
    w = (torch.randn(1, 64, 1, 1, generator=g) * 0.2).requires_grad_(True)
    bias = torch.tensor([0.1], requires_grad=True)
    depth = torch.rand(b, 1, 2 * h, 2 * h, generator=g)
    dw_, db_ = torch.tensor([1.1], requires_grad=True), torch.tensor([-0.2], requires_grad=True)
    ref = po.aux_head(a1, w, bias)
    if use_depth:
        ref = ref * po.depth_head(depth, dw_, db_)
    n = (h // 2) ** 2
    ld = n + 8
    dout = torch.randn(b, n, generator=g)
    grads = torch.autograd.grad(ref, (a1, w, bias) + ((dw_, db_) if use_depth else ()), dout)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    a1d = nhwc(a1.detach()).to(dtype).to(DEV)
    wd, bd = w.detach().reshape(64).to(DEV), bias.detach().to(DEV)
    feat = xhat = None
    if use_depth:
        feat = torch.empty(b, n, device=DEV)
        xhat = torch.empty(b, n, device=DEV)
        depth_d, dw_d, db_d = depth.to(DEV), dw_.detach().to(DEV), db_.detach().to(DEV)  # keep alive past the launch
        lib.rpe_depth_head_fwd(P(depth_d), P(dw_d), P(db_d), P(feat), P(xhat), b, 2 * h, 2 * h, S)
        assert rel_err(feat, po.depth_head(depth, dw_, db_)) < 1e-4
    out = torch.zeros(b, ld, device=DEV)
    raw = torch.empty(b, n, device=DEV)
    idx = torch.empty(b, n, dtype=torch.uint8, device=DEV)
    code = ops.dtype_code(dtype)
    lib.rpe_aux_head_fwd(code, P(a1d), P(wd), P(bd), None if feat is None else P(feat), P(out), ld, P(raw), P(idx), b, h, h, S)
    assert rel_err(out[:, :n], ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    doutd = torch.zeros(b, ld, device=DEV)
    doutd[:, :n] = dout.to(DEV)
    d_a1 = torch.empty_like(a1d)
    gw, gb = torch.zeros(64, device=DEV), torch.zeros(1, device=DEV)
    d_feat = torch.empty(b, n, device=DEV) if use_depth else None
    lib.rpe_aux_head_bwd(code, P(doutd), ld, P(a1d), P(wd), None if feat is None else P(feat), P(raw), P(idx), P(d_a1), P(gw), P(gb),
                         None if d_feat is None else P(d_feat), b, h, h, S)
    t = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(d_a1), grads[0]) < t
    assert rel_err(gw, grads[1].reshape(64)) < t and rel_err(gb, grads[2]) < t
    # the deterministic form: per-block partial sums through a workspace, overwriting; repeats bitwise
    ws = torch.empty(lib.rpe_aux_head_bwd_workspace_floats(code, b, h, h), device=DEV)
    gw2, gb2 = torch.full((64,), 7.0, device=DEV), torch.full((1,), 7.0, device=DEV)
    args = (code, P(doutd), ld, P(a1d), P(wd), None if feat is None else P(feat), P(raw), P(idx), None, P(gw2), P(gb2),
            None if d_feat is None else P(d_feat), b, h, h, P(ws), ws.numel(), S)
    lib.rpe_aux_head_bwd_det(*args)
    assert rel_err(gw2, grads[1].reshape(64)) < t and rel_err(gb2, grads[2]) < t
    first = (gw2.clone(), gb2.clone())
    lib.rpe_aux_head_bwd_det(*args)
    assert torch.equal(gw2, first[0]) and torch.equal(gb2, first[1])
    if use_depth:
        gdw, gdb = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
        lib.rpe_depth_head_bwd(P(d_feat), P(xhat), b * n, P(gdw), P(gdb), S)
        assert rel_err(gdw, grads[3]) < t and rel_err(gdb, grads[4]) < t


def test_lstm_cell_matches_oracle():
    import ctypes
    from rgb_proprioceptive_pose_estimator_amd._lib import lib
    g = torch.Generator().manual_seed(2)
    S_, N, I_, H = 3, 5, 12, 16
    x = torch.randn(S_, N, I_, generator=g)
    w_ih = (torch.randn(4 * H, I_, generator=g) * 0.3).requires_grad_(True)
    w_hh = (torch.randn(4 * H, H, generator=g) * 0.3).requires_grad_(True)
    b_ih = (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True)
    b_hh = (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True)
    out_ref, _, _ = po.lstm_forward(x, w_ih, w_hh, b_ih, b_hh)
    dout = torch.randn(out_ref.shape, generator=g)
    g_wih, g_whh, g_bih, g_bhh = torch.autograd.grad(out_ref, (w_ih, w_hh, b_ih, b_hh), dout)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = lambda t: t.detach().to(DEV).contiguous()
    xd, wih, whh, bih, bhh = d(x), d(w_ih), d(w_hh), d(b_ih), d(b_hh)
    xg = ops.linear_fwd(xd.view(S_ * N, I_), wih).view(S_, N, 4 * H)
    gates = torch.empty(S_, N, 4 * H, device=DEV)
    hs = torch.empty(S_, N, H, device=DEV)
    cs = torch.empty(S_, N, H, device=DEV)
    for t in range(S_):
        if t == 0:
            gates[t].copy_(xg[t])
        else:
            ops.linear_fwd(hs[t - 1], whh, addend=xg[t], out=gates[t])
        lib.rpe_lstm_cell_fwd(P(gates[t]), P(bih), P(bhh), P(cs[t - 1]) if t else None, P(cs[t]), P(hs[t]), N, H, S)
    assert rel_err(hs, out_ref) < 1e-5
    # backward through time
    dgates = torch.empty(S_, N, 4 * H, device=DEV)
    dc = torch.zeros(N, H, device=DEV)
    dh_rec = torch.zeros(N, H, device=DEV)
    whh_t = ops.transpose_f32(whh)  # [H, 4H]
    doutd = d(dout)
    for t in reversed(range(S_)):
        dh = doutd[t] + dh_rec
        lib.rpe_lstm_cell_bwd(P(gates[t]), P(cs[t - 1]) if t else None, P(cs[t]), P(dh), P(dc), P(dgates[t]), N, H, S)
        if t:
            dh_rec = ops.linear_fwd(dgates[t], whh_t, n=H, k=4 * H).contiguous()
    dg2 = dgates.view(S_ * N, 4 * H)
    gw_ih = torch.zeros(4 * H, I_, device=DEV)
    ops.linear_wgrad(dg2, xd.view(S_ * N, I_), gw_ih)
    gw_hh = torch.zeros(4 * H, H, device=DEV)
    ops.linear_wgrad(dg2[N:], hs[:-1].reshape((S_ - 1) * N, H), gw_hh)
    gb = ops.colsum(dg2)
    assert rel_err(gw_ih, g_wih) < 1e-4 and rel_err(gw_hh, g_whh) < 1e-4
    assert rel_err(gb, g_bih) < 1e-4 and rel_err(gb, g_bhh) < 1e-4


METRICS = {"l2": 0, "l1": 1, "linf": 2, "combined": 3}


def test_pose_loss_matches_golden_and_oracle(golden_dir):
    gold = np.load(os.path.join(golden_dir, "pose_loss.npz"))
    pred, truth = torch.from_numpy(gold["pred"]), torch.from_numpy(gold["truth"])
    for metric, mi in METRICS.items():
        for mode, mo in (("position", 0), ("pose", 1)):
            for scale, alpha in ((1.0, 1.0), (2.5, 0.5)):
                out3, grad = ops.pose_loss(pred.to(DEV), truth.to(DEV), mi, mo, scale, alpha, 1e-4)
                tag = "%s_%s_%g_%g" % (metric, mode, scale, alpha)
                np.testing.assert_allclose(out3[0].item(), gold["loss::" + tag], rtol=1e-5)
                np.testing.assert_allclose(grad.cpu().numpy(), gold["grad::" + tag], rtol=1e-4, atol=1e-6)
    out3, _ = ops.pose_loss(pred.to(DEV), truth.to(DEV), 0, 1, 1.0, 1.0, 1e-4, want_grad=False)
    np.testing.assert_allclose(out3[1].item(), gold["val_pos"], rtol=1e-5)
    np.testing.assert_allclose(out3[2].item(), gold["val_ori"], rtol=1e-5)
    # a larger seeded batch against the oracle
    b = po.synth_batch((64, 4), 77)
    p = (b["obj"] + 0.3 * torch.randn(64, 4, 7, generator=torch.Generator().manual_seed(1))).requires_grad_(True)
    ref = po.pose_loss(p, b["x0"], "combined", 1.0, 0.5, 1e-4, "pose")
    (gref,) = torch.autograd.grad(ref, p)
    out3, grad = ops.pose_loss(p.detach().to(DEV), b["x0"].to(DEV), 3, 1, 1.0, 0.5, 1e-4)
    np.testing.assert_allclose(out3[0].item(), ref.item(), rtol=1e-5)
    assert rel_err(grad, gref) < 1e-5
    pe, oe = po.pose_loss(p.detach(), b["x0"], mode="val")
    np.testing.assert_allclose(out3[1].item(), float(pe), rtol=1e-5)
    np.testing.assert_allclose(out3[2].item(), oe, rtol=1e-4)


def test_adam_matches_oracle():
    g = torch.Generator().manual_seed(4)
    n = 10007
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    for step in (1, 2, 3):
        gstep = gr * step
        po.adam_update(p, gstep, m, v, step, lr=1e-3)
        ops.adam_step(pd, gstep.to(DEV), md, vd, 1e-3, 0.9, 0.999, 1e-8, step)
    assert rel_err(pd, p) < 1e-6 and rel_err(md, m) < 1e-6 and rel_err(vd, v) < 1e-6


def test_small_utils():
    g = torch.Generator().manual_seed(8)
    w = torch.randn(37, 50, generator=g)
    wt = ops.transpose_f32(w.to(DEV))
    assert torch.equal(wt[:, :37].cpu(), w.t()) and (wt[:, 37:] == 0).all()
    out, dy = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.equal(ops.relu_bwd(out.to(DEV), dy.to(DEV)).cpu(), dy * (out > 0))
    x = torch.randn(300, 70, generator=g)
    assert rel_err(ops.colsum(x.to(DEV)), x.sum(0)) < 1e-5
    dst = torch.zeros(300, 80, device=DEV)
    ops.copy2d(x.to(DEV), dst[:, 5:], cols=70)
    assert torch.equal(dst[:, 5:75].cpu(), x)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_aux,use_depth", [(False, False), (True, False), (True, True)])
def test_stem_backward_fused(dtype, use_aux, use_depth):
    """rpe_stem_bwd == autograd through relu(bn1(y)) -> {maxpool 3x3/2, aux conv 64->1 + maxpool 2 (x depth feature)}."""
    import ctypes
    from rgb_proprioceptive_pose_estimator_amd._lib import lib
    g = torch.Generator().manual_seed(21)
    b, h = 3, 12
    y = q(torch.randn(b, 64, h, h, generator=g) * 1.5 + 0.2, dtype).requires_grad_(True)
    gamma = (torch.rand(64, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(64, generator=g) * 0.3).requires_grad_(True)
    mean = y.detach().mean((0, 2, 3))
    var = y.detach().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma.detach() * invstd, beta.detach() - mean * gamma.detach() * invstd
    a1 = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    pool = F.max_pool2d(a1, 3, 2, 1)
    dpool = q(torch.randn(pool.shape, generator=g), dtype)
    loss = (pool * dpool).sum()
    w = torch.randn(1, 64, 1, 1, generator=g) * 0.2
    bias = torch.tensor([0.1])
    n = (h // 2) ** 2
    ld = n + 4
    dout = torch.randn(b, n, generator=g)
    df = torch.rand(b, n, generator=g) + 0.5
    if use_aux:
        aux = po.aux_head(a1, w, bias)
        if use_depth:
            aux = aux * df
        loss = loss + (aux * dout).sum()
    gy, gg, gb_ = torch.autograd.grad(loss, (y, gamma, beta))

    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = ops.dtype_code(dtype)
    yd = nhwc(y.detach()).to(dtype).to(DEV)
    sc, sh, mu, iv, gm = (t.float().to(DEV) for t in (scale, shift, mean, invstd, gamma.detach()))
    a1d = ops.bn_apply(yd, sc, sh, None, True)
    _, pidx = ops.maxpool_fwd(a1d)
    dpd = nhwc(dpool).to(dtype).to(DEV)
    aux_idx = aux_w = doutd = dfd = None
    if use_aux:
        aux_w, bd = w.reshape(64).to(DEV), bias.to(DEV)
        dfd = df.to(DEV) if use_depth else None
        out = torch.zeros(b, ld, device=DEV)
        raw = torch.empty(b, n, device=DEV)
        aux_idx = torch.empty(b, n, dtype=torch.uint8, device=DEV)
        lib.rpe_aux_head_fwd(code, P(a1d), P(aux_w), P(bd), P(dfd), P(out), ld, P(raw), P(aux_idx), b, h, h, S)
        doutd = torch.zeros(b, ld, device=DEV)
        doutd[:, :n] = dout.to(DEV)
    dgam, dbet = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    dy = torch.empty_like(yd)
    part = torch.empty(2 * 1024 * 64, device=DEV)
    c1c2 = torch.empty(128, device=DEV)
    dpart = torch.zeros(256 * 2 * 64 + 64, dtype=torch.float64, device=DEV)
    lib.rpe_stem_bwd(code, P(dpd), P(pidx), P(yd), P(sc), P(sh), P(mu), P(iv), P(gm), P(doutd), ld, P(dfd), P(aux_idx), P(aux_w), P(dgam), P(dbet),
                     P(dy), b, h, h, P(part), part.numel(), P(c1c2), P(dpart), S)
    t = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(nchw(dy), gy) < t
    assert rel_err(dgam, gg) < t and rel_err(dbet, gb_) < t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 14, 64, 64, 3, 1, 1), (2, 14, 256, 512, 1, 2, 0), (3, 8, 64, 256, 1, 1, 0), (8, 24, 128, 128, 3, 1, 1)])
@pytest.mark.parametrize("residual", [False, True])
def test_conv_fwd_affine_inference_form(dtype, cfg, residual):
    """relu(conv(x, w*scale) + shift (+ identity)) in one launch == conv -> BatchNorm2d(eval) -> add -> ReLU."""
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    g = torch.Generator().manual_seed(5)
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.3
    wq = q(w * scale[:, None, None, None], dtype)
    ref = F.conv2d(x, wq, None, s, p) + shift[None, :, None, None]
    add = None
    if residual:
        add = q(torch.randn(ref.shape, generator=g), dtype)
        ref = ref + add
    ref = F.relu(ref)
    out = ops.conv2d_fwd_affine(nhwc(x).to(dtype).to(DEV), wq.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), s, p, shift.to(DEV),
                                None if add is None else nhwc(add).to(dtype).to(DEV), True)
    assert rel_err(nchw(out), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c", [(2 * 14 * 14, 256), (3 * 7 * 7, 2048), (1000, 64)])
def test_bn_apply_with_shortcut_bn_on_the_fly(dtype, rows, c):
    """rpe_bn_apply_res_bn == BN apply of the projection shortcut (rounded to the compute dtype) followed by the residual BN apply,
    up to that one rounding; the packed mask marks exactly the positive outputs."""
    g = torch.Generator().manual_seed(rows + c)
    y, ry = q(torch.randn(rows, c, generator=g), dtype), q(torch.randn(rows, c, generator=g) * 2 + 0.5, dtype)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rsc, rsh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    ref = F.relu(y * sc + sh + ry * rsc + rsh)
    yd, rd = y.to(dtype).to(DEV), ry.to(dtype).to(DEV)
    out = ops.bn_apply_res_bn(yd, sc.to(DEV), sh.to(DEV), rd, rsc.to(DEV), rsh.to(DEV))
    assert rel_err(out, ref) < tol(dtype)
    two = ops.bn_apply(yd, sc.to(DEV), sh.to(DEV), ops.bn_apply(rd, rsc.to(DEV), rsh.to(DEV), None, relu=False), relu=True)
    assert rel_err(out, two) < tol(dtype)
    norelu = ops.bn_apply_res_bn(yd, sc.to(DEV), sh.to(DEV), rd, rsc.to(DEV), rsh.to(DEV), relu=False)
    assert rel_err(norelu, y * sc + sh + ry * rsc + rsh) < tol(dtype)
    if dtype != torch.float32:
        out2, mask = ops.bn_apply_res_bn(yd, sc.to(DEV), sh.to(DEV), rd, rsc.to(DEV), rsh.to(DEV), want_mask=True)
        assert torch.equal(out2, out)
        bits = ((mask.cpu()[:, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(rows, c).bool()
        assert torch.equal(bits, out.cpu().float() > 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(8, 28, 64, 256), (4, 14, 128, 512), (6, 14, 256, 1024), (2, 6, 64, 256), (3, 30, 64, 256), (5, 14, 128, 512), (33, 28, 64, 512)])
@pytest.mark.parametrize("shortcut", ["identity", "projection", "none"])
def test_conv1x1_forward_with_bn_from_gram(dtype, cfg, shortcut):
    """conv3 -> bn3 -> (+identity) -> ReLU without writing or re-reading the conv output: BatchNorm statistics from the Gram matrix of
    the INPUT (rpe_gram + rpe_bn_stats_from_gram) against the statistics of the conv output itself; the fused forward
    (rpe_conv1x1_fwd_bn) against conv -> batch_norm -> add -> relu; mask = the positive outputs."""
    b, h, ci, co = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = q(F.relu(torch.randn(b, h, h, ci, generator=g) * 1.5 + 0.4), dtype)          # post-ReLU activations: positive means
    w = q(torch.randn(co, ci, generator=g) / ci ** 0.5, dtype)
    gamma, beta = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.3
    rows = b * h * h
    y64 = x.reshape(rows, ci).double() @ w.double().t()
    mean_ref, var_ref = y64.mean(0), y64.var(0, unbiased=False)
    xd, wd = x.to(dtype).to(DEV), w.to(dtype).to(DEV)
    S, s1, buf = ops.gram(xd)
    assert rel_err(S, (x.reshape(rows, ci).double().t() @ x.reshape(rows, ci).double()).float()) < 2e-5
    assert rel_err(s1, x.reshape(rows, ci).double().sum(0).float()) < 2e-5
    assert torch.equal(buf, ops.gram(xd)[2])                                          # fixed-order sums
    rm, rv = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)
    scale, shift, mean, invstd = ops.bn_stats_from_gram(wd, buf, rows, gamma.to(DEV), beta.to(DEV), rm, rv)
    assert rel_err(mean, mean_ref.float()) < 1e-4
    assert ((invstd.cpu().double() * torch.sqrt(var_ref + 1e-5) - 1).abs().max()) < 2e-4
    assert rel_err(rm, 0.1 * mean_ref.float()) < 1e-4
    assert rel_err(rv, (0.9 + 0.1 * var_ref * rows / (rows - 1)).float()) < 2e-4
    res = rs = rb = None
    ref = (y64 - mean_ref) / torch.sqrt(var_ref + 1e-5) * gamma.double() + beta.double()
    if shortcut != "none":
        res = q(torch.randn(rows, co, generator=g), dtype)
        if shortcut == "projection":
            rs, rb = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.3
            ref = ref + res.double() * rs.double() + rb.double()
        else:
            ref = ref + res.double()
    ref = F.relu(ref).float().reshape(b, h, h, co)
    dev = lambda t: None if t is None else t.to(DEV)
    out, mask, y = ops.conv1x1_fwd_bn(xd, wd, scale, shift, None if res is None else res.to(dtype).to(DEV).reshape(b, h, h, co), dev(rs), dev(rb), want_y=True)
    assert rel_err(out, ref) < tol(dtype)
    assert rel_err(y, y64.float().reshape(b, h, h, co)) < tol(dtype)
    bits = ((mask.cpu()[:, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(rows, co).bool()
    assert torch.equal(bits, out.cpu().float().reshape(rows, co) > 0)
    # without y the 64 / 128 -> 256 k launches of >= 512 rows take the row-streaming kernel (csrc/stream1x1.hip; ragged spans in the (3, 30, ..)
    # and (5, 14, ..) cases): bitwise the tiled form, in both walk directions
    out2, mask2, none = ops.conv1x1_fwd_bn(xd, wd, scale, shift, None if res is None else res.to(dtype).to(DEV).reshape(b, h, h, co), dev(rs), dev(rb))
    assert none is None and torch.equal(out2, out) and torch.equal(mask2, mask)
    if rows >= 512 and ci in (64, 128) and co % 256 == 0 and shortcut != "none":
        assert ops.last_kernel_name().startswith("conv1x1_stream_fwd_kernel")
    ops.set_walk_direction(1)
    try:
        out3, mask3, _ = ops.conv1x1_fwd_bn(xd, wd, scale, shift, None if res is None else res.to(dtype).to(DEV).reshape(b, h, h, co), dev(rs), dev(rb))
    finally:
        ops.set_walk_direction(0)
    assert torch.equal(out3, out) and torch.equal(mask3, mask)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,c", [(8 * 28 * 28, 64), (3 * 14 * 14, 128), (72, 64), (200, 128), (40000, 64), (70001, 128)])
def test_bn_apply_fused_with_gram(dtype, rows, c):
    """rpe_bn_apply_gram: bitwise the output of rpe_bn_apply (ReLU), and the Gram matrix / column sums of THAT output (what rpe_gram
    computes from it in a second pass) -- any row count incl. ragged tails and several workgroups; repeatable bit for bit."""
    g = torch.Generator().manual_seed(rows + c)
    y = q(torch.randn(rows, c, generator=g) * 1.5 + 0.3, dtype)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    yd = y.to(dtype).to(DEV)
    out, S, s1, buf = ops.bn_apply_gram(yd, sc.to(DEV), sh.to(DEV))
    ref = ops.bn_apply(yd, sc.to(DEV), sh.to(DEV), None, relu=True)
    assert torch.equal(out, ref)
    a = out.double().cpu()
    assert rel_err(S, (a.t() @ a).float()) < 2e-5
    assert rel_err(s1, a.sum(0).float()) < 2e-5
    S2, s12, _ = ops.gram(out)
    assert rel_err(S, S2) < 2e-5 and rel_err(s1, s12) < 2e-5
    assert torch.equal(buf, ops.bn_apply_gram(yd, sc.to(DEV), sh.to(DEV))[3])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(4, 28, 64, 256, 64), (3, 14, 128, 512, 128), (2, 6, 64, 256, 128), (2, 14, 256, 1024, 256), (40, 56, 64, 256, 64),
                                 (37, 15, 128, 512, 256)])
def test_y3_free_bottleneck_backward(dtype, cfg):
    """The backward of a block whose raw conv3 output was never written.  a2 -> conv3 (1x1) -> bn3 -> + identity -> ReLU = a3 -> next
    conv1 (1x1).  (a) the next block's fused conv1 data gradient with bn->y = NULL: the same dz as with y, sum dz in the partial rows
    and zeros in their second half; (b) rpe_bn_backward_coeffs_t (sum dz*xhat from T = dz^T a2 and W) == the coefficients from y itself,
    dgamma / dbeta == autograd; (c) rpe_conv1x1_wgrad_combine (T + the forward's Gram buffer) == autograd's conv3 weight gradient;
    (d) the folded data gradient from those coefficients == autograd's gradient of a2."""
    b, h, p, co, nxt = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    rows = b * h * h
    a2 = q(F.relu(torch.randn(rows, p, generator=g) * 1.2 + 0.3), dtype)
    w3 = q(torch.randn(co, p, generator=g) / p ** 0.5, dtype)
    idn = q(F.relu(torch.randn(rows, co, generator=g)), dtype)
    gamma, beta = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.3
    w1n = q(torch.randn(nxt, co, generator=g) / co ** 0.5, dtype)          # the next block's conv1 [out, in]
    dy1 = q(torch.randn(rows, nxt, generator=g), dtype)                     # gradient of its raw output
    sc_grad = q(torch.randn(rows, co, generator=g), dtype)                 # the shortcut's gradient term
    # GPU forward pieces
    dev = lambda t: t.to(DEV)
    a2d, w3d = a2.to(dtype).to(DEV).reshape(b, h, h, p), w3.to(dtype).to(DEV)
    S, s1, buf = ops.gram(a2d)
    scale, shift, mean, invstd = ops.bn_stats_from_gram(w3d, buf, rows, dev(gamma), dev(beta))
    out, mask, y = ops.conv1x1_fwd_bn(a2d, w3d, scale, shift, idn.to(dtype).to(DEV).reshape(b, h, h, co), want_y=True)
    bits = ((mask.cpu()[:, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(rows, co).double()
    # reference in double; the ReLU gate is the forward's packed mask (elements within rounding of 0 may fall on either side)
    a2r = a2.double().requires_grad_(True)
    w3r = w3.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y3 = a2r @ w3r.t()
    z = F.batch_norm(y3, None, None, gr, br, True, 0.1, 1e-5)
    pre = z + idn.double()
    assert ((pre.detach() > 0).double() != bits).double().mean() < 2e-3
    a3 = pre * bits
    dA = dy1.double() @ w1n.double() + sc_grad.double()
    da2_ref, dw_ref, dg_ref, db_ref = torch.autograd.grad(a3, (a2r, w3r, gr, br), dA)
    dz_ref = dA * bits
    # (a)
    w_crsk = w1n.t().contiguous().reshape(co, 1, 1, nxt).to(dtype).to(DEV)
    dyd, add = dy1.to(dtype).to(DEV).reshape(b, h, h, nxt), sc_grad.to(dtype).to(DEV).reshape(b, h, h, co)
    t = {torch.bfloat16: 3e-2, torch.float16: 5e-3}[dtype]
    dz_y, st_y = ops.conv2d_dgrad_bn(dyd, w_crsk, (b, h, h, co), 1, 0, y, mean, invstd, addend=add, a_mask=mask)
    dz, st = ops.conv2d_dgrad_bn(dyd, w_crsk, (b, h, h, co), 1, 0, None, None, None, addend=add, a_mask=mask)
    assert torch.equal(dz, dz_y) and float(st[:, 1].abs().max()) == 0.0
    assert rel_err(st[:, 0].sum(0), dz.float().reshape(rows, co).sum(0)) < 1e-5          # the sum of dz AS STORED (rounded), so that it is
    assert rel_err(st[:, 0].sum(0), st_y[:, 0].sum(0)) < t                              # consistent with T = dz^T a2 in (b)
    assert rel_err(dz.reshape(rows, co), dz_ref.float()) < t
    # (a') the same launch with the T side product: identical dz and partial sums, T == the separate dz^T a2 launch (up to the fp32
    # summation order), repeatable bit for bit
    if p in (64, 128):
        dz_t, st_t, T_t = ops.conv1x1_dgrad_bn_t(dyd, w_crsk, (b, h, h, co), a2d, mask, addend=add)
        assert torch.equal(dz_t, dz) and torch.equal(st_t, st)
        assert rel_err(T_t, ops.conv2d_wgrad(a2d, dz, 1, 1, 0).reshape(co, p)) < 2e-5
        assert rel_err(T_t, (dz.double().cpu().reshape(rows, co).t() @ a2.double()).float()) < 2e-5
        assert torch.equal(T_t, ops.conv1x1_dgrad_bn_t(dyd, w_crsk, (b, h, h, co), a2d, mask, addend=add)[2])
    # (b)
    T = ops.conv2d_wgrad(a2d, dz, 1, 1, 0).reshape(co, p)
    dg, db, c1c2 = ops.bn_backward_coeffs_t(st, rows, T, w3d, mean, invstd)
    dg_y, db_y, c1c2_y = ops.bn_backward_coeffs(st_y, rows)
    assert rel_err(db, db_y) < t and rel_err(c1c2[0], c1c2_y[0]) < t
    dzf = dz.float().cpu().reshape(rows, co).double()
    y_exact = a2.double() @ w3.double().t()
    q_exact = ((y_exact - mean.cpu().double()) * invstd.cpu().double() * dzf).sum(0)
    assert rel_err(dg, q_exact.float()) < 1e-4                              # the T form sees the unrounded y and the stored dz
    assert rel_err(dg, dg_ref.float()) < t and rel_err(db, db_ref.float()) < t
    # (c)
    w3m = w3.to(DEV).contiguous()
    dw = ops.conv1x1_wgrad_combine(T, buf, w3m, dev(gamma), invstd, mean, c1c2)
    assert rel_err(dw, dw_ref.float()) < t
    assert torch.equal(dw, ops.conv1x1_wgrad_combine(T, buf, w3m, dev(gamma), invstd, mean, c1c2))
    # (d)
    wk, bias = ops.bn_bwd_fold_conv1x1(w3d, w3d.t().contiguous(), dev(gamma), invstd, mean, c1c2)
    dx = ops.conv1x1_dgrad_kcat(dz, a2d, wk, bias)
    assert rel_err(dx.reshape(rows, p), da2_ref.float()) < t


def test_operands_above_2_gib():
    """Round 1 capped a tensor at 2 GiB (32-bit byte offsets against one buffer descriptor: ~1300 images per process in bf16).  The
    descriptors now start at each tile's / split's first row or image: convs whose activation tensors exceed 2 GiB must equal the same
    convs run on their two halves (forward, data gradient, weight gradient; 1x1 dense and 3x3 gathered)."""
    dtype = torch.bfloat16
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(3)
    b, h, ci, co = 1400, 28, 1024, 64                                  # 1x1: x [1400,28,28,1024] = 2.25 GB in, 64 channels out
    x = torch.randn(b // 2, h, h, ci, generator=g).to(dtype)
    x = torch.cat([x, x.flip(0)], 0).to(DEV)                          # 2.25 GB on the device
    assert x.numel() * 2 > (1 << 31)
    w = (torch.randn(co, 1, 1, ci, generator=g) / ci ** 0.5).to(dtype).to(DEV)
    y = ops.conv2d_fwd(x, w, 1, 0)
    hb = b // 2
    y0 = ops.conv2d_fwd(x[:hb].contiguous(), w, 1, 0)
    y1 = ops.conv2d_fwd(x[hb:].contiguous(), w, 1, 0)
    assert torch.equal(y[:hb], y0) and torch.equal(y[hb:], y1)
    # weight gradient reading the > 2 GiB activation tensor (dense TN): equals the sum of the halves' gradients
    dy = torch.randn(b, h, h, co, generator=g).to(dtype).to(DEV)
    dw = ops.conv2d_wgrad(x, dy, 1, 1, 0)
    dwh = ops.conv2d_wgrad(x[:hb].contiguous(), dy[:hb].contiguous(), 1, 1, 0) + ops.conv2d_wgrad(x[hb:].contiguous(), dy[hb:].contiguous(), 1, 1, 0)
    assert rel_err(dw, dwh) < 1e-3
    # data gradient WRITING a > 2 GiB tensor from a small dy (dense NT, role 1)
    wd = w.reshape(co, ci).t().contiguous().reshape(ci, 1, 1, co)
    dx = ops.conv2d_dgrad(dy, wd, tuple(x.shape), 1, 0)
    dx0 = ops.conv2d_dgrad(dy[:hb].contiguous(), wd, (hb, h, h, ci), 1, 0)
    assert torch.equal(dx[:hb], dx0)
    assert torch.equal(dx[hb:], ops.conv2d_dgrad(dy[hb:].contiguous(), wd, (hb, h, h, ci), 1, 0))
    del dx, dx0, x, y, y0, y1, dw, dwh
    torch.cuda.empty_cache()
    # 3x3 gathered forward + weight gradient over a > 2 GiB input: [1400,56,56,256] = 2.25 GB
    ci3, co3, h3 = 256, 64, 56
    x3 = torch.randn(b // 2, h3, h3, ci3, generator=g).to(dtype)
    x3 = torch.cat([x3, x3.flip(0)], 0).to(DEV)
    assert x3.numel() * 2 > (1 << 31)
    w3 = (torch.randn(co3, 3, 3, ci3, generator=g) / (9 * ci3) ** 0.5).to(dtype).to(DEV)
    y3 = ops.conv2d_fwd(x3, w3, 1, 1)
    assert torch.equal(y3[:hb], ops.conv2d_fwd(x3[:hb].contiguous(), w3, 1, 1))
    assert torch.equal(y3[hb:], ops.conv2d_fwd(x3[hb:].contiguous(), w3, 1, 1))
    dy3 = torch.randn(b, h3, h3, co3, generator=g).to(dtype).to(DEV)
    dw3 = ops.conv2d_wgrad(x3, dy3, 3, 1, 1)
    dw3h = ops.conv2d_wgrad(x3[:hb].contiguous(), dy3[:hb].contiguous(), 3, 1, 1) + ops.conv2d_wgrad(x3[hb:].contiguous(), dy3[hb:].contiguous(), 3, 1, 1)
    assert rel_err(dw3, dw3h) < 1e-3


SPLITK = [  # rollout-frame shapes (one image through layer2-4): few 64x64 output tiles, long reductions
    (1, 7, 512, 512, 3, 1, 1), (1, 14, 256, 256, 3, 1, 1), (1, 28, 128, 128, 3, 1, 1), (1, 7, 2048, 512, 1, 1, 0), (1, 14, 1024, 2048, 1, 2, 0),
    (1, 14, 512, 512, 3, 2, 1), (3, 7, 512, 512, 3, 1, 1), (2, 9, 320, 72, 3, 1, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", SPLITK)
def test_conv_fwd_affine_split_k(dtype, cfg):
    """The inference forward splits K over grid.y when a launch has few output tiles (rpe_conv2d_fwd_affine_ws): the planned
    split must be taken, match the conv -> BN(eval) -> add -> ReLU reference and the unsplit launch, and repeat bitwise."""
    import ctypes
    from rgb_proprioceptive_pose_estimator_amd._lib import lib
    b, h, ci, co, k, s, p = cfg
    x, w = _conv_inputs(cfg, dtype)
    g = torch.Generator().manual_seed(7)
    shift = torch.randn(co, generator=g) * 0.3
    wq = q(w, dtype)
    ref = F.conv2d(x, wq, None, s, p) + shift[None, :, None, None]
    add = q(torch.randn(ref.shape, generator=g), dtype)
    ref = F.relu(ref + add)
    xd, wd, ad = nhwc(x).to(dtype).to(DEV), wq.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), nhwc(add).to(dtype).to(DEV)
    d = ops.conv_desc(xd.shape, co, k, s, p)
    assert lib.rpe_conv2d_fwd_affine_workspace_bytes(ctypes.byref(d), ops.dtype_code(xd)) > 0, "this shape is expected to split"
    out = ops.conv2d_fwd_affine(xd, wd, s, p, shift.to(DEV), ad, True)
    assert "nt_split_epilogue_kernel" in ops.last_kernel_name()
    assert rel_err(nchw(out), ref) < tol(dtype)
    plain = ops.conv2d_fwd_affine(xd, wd, s, p, shift.to(DEV), ad, True, split_k=False)
    assert "nt_split_epilogue_kernel" not in ops.last_kernel_name()
    assert rel_err(out, plain) < (2e-6 if dtype == torch.float32 else tol(dtype))
    for _ in range(3):
        assert torch.equal(ops.conv2d_fwd_affine(xd, wd, s, p, shift.to(DEV), ad, True), out)
    nob = ops.conv2d_fwd_affine(xd, wd, s, p, shift.to(DEV), None, False)      # bias only
    assert rel_err(nchw(nob), F.conv2d(x, wq, None, s, p) + shift[None, :, None, None]) < tol(dtype)


def test_bn_reduce_handoff_under_load():
    """The multi-slice reduce+finalize launch hands partial sums between workgroups with write-through stores, a drained
    counter add and L1-bypassing loads instead of release/acquire fences (csrc/norm.hip).  Hammer it while another stream
    streams through HBM: every repetition must reproduce the first one bitwise and match a double-precision reference."""
    g = torch.Generator().manual_seed(0)
    tiles, c, rows = 6272, 256, 6272 * 128
    part = (torch.randn(tiles, 2, c, generator=g) * 3 + 1).to(DEV)
    part[:, 1] = part[:, 1].abs() * 40 + 130          # sum of squares large enough for a positive variance
    gamma, beta = torch.rand(c, generator=g).to(DEV) + 0.5, torch.rand(c, generator=g).to(DEV)
    mean_ref = part[:, 0].double().sum(0) / rows
    var_ref = part[:, 1].double().sum(0) / rows - mean_ref * mean_ref
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(1 << 28, dtype=torch.uint8, device=DEV), torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    first = None
    for it in range(60):
        with torch.cuda.stream(side):
            for _ in range(3):
                big_b.copy_(big_a)                      # ~1.5 GB of concurrent HBM traffic per repetition
        scale, shift, mean, invstd = ops.bn_finalize(part, rows, gamma, beta)
        res = torch.stack([scale, shift, mean, invstd])
        if first is None:
            first = res.clone()
            assert rel_err(mean, mean_ref.float()) < 1e-6
            assert rel_err(invstd, (1.0 / torch.sqrt(var_ref + 1e-5)).float()) < 1e-5
        else:
            assert torch.equal(res, first), "repetition %d differs" % it
    torch.cuda.synchronize()


# ------------------------------------------------------------------ torch.ops.rpe.* (dispatcher registration, torch_ops.py)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 64, 3, 1, 1), (2, 16, 16, 64, 128, 3, 2, 1), (2, 28, 28, 64, 256, 1, 1, 0)])
def test_torch_ops_conv2d_autograd(dtype, cfg):
    """`torch.ops.rpe.conv2d` (one differentiable dispatcher op built from the forward / data-gradient / weight-gradient launches) against
    F.conv2d and torch autograd on the CPU in fp32; the plain ops are bitwise the ctypes wrappers they register."""
    import rgb_proprioceptive_pose_estimator_amd.torch_ops  # noqa: F401  (registers torch.ops.rpe)
    b, h, w_, ci, co, k, s, p = cfg
    g = torch.Generator().manual_seed(b * 100 + co + k)
    x = torch.randn(b, ci, h, w_, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    xr, wr = q(x, dtype).clone().requires_grad_(), q(w, dtype).clone().requires_grad_()   # (q() may return its argument itself)
    yr = F.conv2d(xr, wr, None, s, p)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(q(dy, dtype))
    xd = nhwc(x).to(dtype).to(DEV).requires_grad_()
    wd = nhwc(w).to(dtype).to(DEV).requires_grad_()
    y = torch.ops.rpe.conv2d(xd, wd, s, p)
    y.backward(nhwc(dy).to(dtype).to(DEV))
    assert rel_err(nchw(y), yr) < tol(dtype)
    assert rel_err(nchw(xd.grad), xr.grad) < tol(dtype) * 2
    assert rel_err(nchw(wd.grad), wr.grad) < max(tol(dtype) * 2, 1e-4)
    assert torch.equal(torch.ops.rpe.conv2d_fwd(xd.detach(), wd.detach(), s, p), ops.conv2d_fwd(xd.detach(), wd.detach(), s, p))
    with pytest.raises(NotImplementedError):   # CUDA (= HIP) dispatch key only: no CPU fallback behind the op
        torch.ops.rpe.conv2d_fwd(nhwc(x), nhwc(w), s, p)


def test_torch_ops_loss_adam_linear_bn():
    """The remaining dispatcher ops: `pose_distance_loss` (value + gradient through autograd) against the oracle's loss, `adam_step`
    against the oracle's update, `linear_fwd` and `bn_apply` against torch on the CPU."""
    import rgb_proprioceptive_pose_estimator_amd.torch_ops  # noqa: F401
    g = torch.Generator().manual_seed(5)
    pred, truth = torch.randn(6, 7, generator=g), torch.randn(6, 7, generator=g)
    truth[:, 3:] = F.normalize(truth[:, 3:], dim=-1)
    pr = pred.clone().requires_grad_()
    ref = po.pose_loss(pr, truth, metric="combined", scale=1.0, alpha=0.5, eps=1e-4, mode="pose")
    ref.backward()
    pd = pred.to(DEV).requires_grad_()
    loss, _ = torch.ops.rpe.pose_distance_loss(pd, truth.to(DEV), 3, 1, 1.0, 0.5, 1e-4)
    (2.0 * loss).backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    assert rel_err(pd.grad, 2.0 * pr.grad) < 1e-5
    # Adam, two steps
    p, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    m, v = torch.zeros(1000), torch.zeros(1000)
    pdv, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    for step in (1, 2):
        po.adam_update(p, gr, m, v, step, lr=1e-3)
        torch.ops.rpe.adam_step(pdv, gr.to(DEV), md, vd, 1e-3, 0.9, 0.999, 1e-8, step)
    assert rel_err(pdv, p) < 1e-6 and rel_err(md, m) < 1e-6 and rel_err(vd, v) < 1e-6
    # Linear + ReLU, BatchNorm affine map + residual + ReLU
    x, w, b = torch.randn(40, 64, generator=g), torch.randn(24, 64, generator=g), torch.randn(24, generator=g)
    assert rel_err(torch.ops.rpe.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), True), F.relu(x @ w.t() + b)) < 2e-5
    y, r = torch.randn(2, 8, 8, 64, generator=g), torch.randn(2, 8, 8, 64, generator=g)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    assert rel_err(torch.ops.rpe.bn_apply(y.to(DEV), sc.to(DEV), sh.to(DEV), r.to(DEV), True), F.relu(y * sc + sh + r)) < 2e-5


# ------------------------------------------------------------------ halo form of the 3x3 weight gradient (csrc/wgrad_halo.hip)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [
    # (B, H, W, Ci, Co): 128-channel form (32-channel chunks) and 64-channel form; several row spans per image, image and tensor borders,
    # maps narrower than the default width bound (min width lowered for the call), H != W, one image, ring wrap-around (56-wide maps)
    (2, 28, 28, 128, 128), (3, 56, 56, 64, 64), (2, 14, 14, 256, 256), (5, 7, 7, 512, 512), (2, 28, 28, 64, 128), (2, 28, 28, 128, 64),
    (1, 20, 36, 192, 128), (9, 12, 9, 64, 64), (1, 56, 56, 128, 256), (4, 60, 60, 64, 64)])
def test_conv_wgrad_halo_form(dtype, cfg):
    """The deterministic 3x3 / stride-1 / pad-1 weight gradient in the halo form (all nine taps in one workgroup over the zero-padded pixel
    grid) against torch's weight gradient on the CPU in fp32; bitwise reproducible; the width bound sends narrow maps back to the gathered
    form and both forms agree to summation-order noise."""
    from rgb_proprioceptive_pose_estimator_amd._lib import lib
    b, h, w_, ci, co = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = q(torch.randn(b, ci, h, w_, generator=g), dtype)
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    y = F.conv2d(x, wt, None, 1, 1)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, wt, dy)
    xd, dyd = nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV)
    prev = lib.rpe_conv2d_wgrad_halo_min_width(2)
    try:
        dw = ops.conv2d_wgrad(xd, dyd, 3, 1, 1)
        assert ops.last_kernel_name().startswith("wgrad_halo"), ops.last_kernel_name()
        assert rel_err(dw.permute(0, 3, 1, 2), ref) < tol(dtype)
        assert torch.equal(dw, ops.conv2d_wgrad(xd, dyd, 3, 1, 1)), "the halo-form weight gradient is not bitwise reproducible"
        lib.rpe_conv2d_wgrad_halo_min_width(1000)
        dwg = ops.conv2d_wgrad(xd, dyd, 3, 1, 1)
        assert not ops.last_kernel_name().startswith("wgrad_halo")
        assert rel_err(dw, dwg) < 1e-4
    finally:
        lib.rpe_conv2d_wgrad_halo_min_width(prev)
