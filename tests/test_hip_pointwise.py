"""Parity of the HBM-bound kernels against torch CPU fp32 primitives, through the C ABI:
norm statistics / norm(+ReLU) backward (instance, batch, group), combine, trilinear x2 and its
adjoint, linear combinations, the entropy objectives, fused Adam, mask + Dice counts.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(name, got, ref, rel=2e-5, abs_=2e-6):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rel * scale + abs_, f"{name}: max|err|={err:.3e} (max|ref|={scale:.3e})"


def cl(x):
    from multimodal_tta_amd import ops
    return ops.to_cl(x.cuda().contiguous())


def ncdhw(x_cl):
    return x_cl.permute(0, 4, 1, 2, 3).contiguous().cpu()


def test_layout_roundtrip():
    from multimodal_tta_amd import ops
    x = torch.randn(2, 3, 5, 6, 7)
    xc = ops.to_cl(x.cuda())
    assert xc.shape == (2, 5, 6, 7, 3) and xc.stride(-1) == 1 and xc.stride(-2) == 4
    assert torch.equal(ncdhw(xc), x)
    back = ops.from_cl(xc)
    assert back.is_contiguous() and torch.equal(back.cpu(), x)


NORM_CASES = [
    ("INSTANCE", 1, (2, 5, 7, 6, 5), False),
    ("INSTANCE", 1, (1, 32, 16, 16, 16), False),
    ("BATCH", 1, (2, 8, 6, 6, 6), True),
    ("GROUP", 4, (2, 8, 6, 6, 6), True),
    ("INSTANCE", 1, (1, 3, 24, 24, 24), False),
]


@pytest.mark.parametrize("kind,groups,shape,affine", NORM_CASES)
def test_norm_forward_backward(kind, groups, shape, affine):
    """y -> relu(norm(y)): statistics from channel_stats partials, applied by combine; backward by the
    reduce / finalize / apply triple; BatchNorm also checks the running-statistics EMA."""
    from multimodal_tta_amd import ops

    torch.manual_seed(3)
    n, c, d, h, w = shape
    y = (torch.randn(shape) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(c) + 0.5).requires_grad_(True) if affine else None
    beta = (torch.randn(c) * 0.1).requires_grad_(True) if affine else None
    rm, rv = torch.zeros(c), torch.ones(c)
    if kind == "INSTANCE":
        ref = F.relu(F.instance_norm(y, eps=1e-5))
    elif kind == "BATCH":
        ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5))
    else:
        ref = F.relu(F.group_norm(y, groups, gamma, beta, eps=1e-5))
    gout = torch.randn_like(ref)
    ref.backward(gout)

    y_cl = cl(y.detach())
    rows = ops.reduce_rows_per_n(y_cl)
    part = torch.empty(n * rows * 2 * c, device="cuda")
    ops.channel_stats(y_cl, part)
    mean = torch.empty(n * c, device="cuda")
    rstd = torch.empty(n * c, device="cuda")
    scratch = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    g_d = gamma.detach().cuda() if affine else None
    b_d = beta.detach().cuda() if affine else None
    rm_d, rv_d = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    k = ops.NORM_KINDS[kind]
    ops.norm_stats_finalize(k, groups, part, rows, n, c, d * h * w, 1e-5, True, rm_d, rv_d, 0.1, mean, rstd, scratch)
    nl = ops.NL(mean, rstd, g_d, b_d, relu=True)
    out = torch.empty_like(y_cl)
    ops.combine(y_cl, nl, None, None, out)
    torch.cuda.synchronize()
    close("norm+relu forward", ncdhw(out), ref)
    if kind == "BATCH":
        close("running_mean", rm_d, rm, rel=1e-5)
        close("running_var", rv_d, rv, rel=1e-5)

    dT = cl(gout)
    bpart = torch.empty(n * rows * 2 * c, device="cuda")
    m1, m2 = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    dg = torch.zeros(c, device="cuda") if affine else None
    db = torch.zeros(c, device="cuda") if affine else None
    ops.norm_bwd_reduce(dT, y_cl, nl, bpart)
    ops.norm_bwd_finalize(k, groups, bpart, rows, n, c, d * h * w, g_d, True, m1, m2, dg, db, False, scratch)
    dy = torch.empty_like(y_cl)
    ops.norm_bwd_apply(dT, y_cl, nl, m1, m2, dy)
    torch.cuda.synchronize()
    close("norm backward dx", ncdhw(dy), y.grad, rel=2e-4, abs_=2e-6)
    if affine:
        close("dgamma", dg, gamma.grad, rel=2e-4, abs_=1e-4)
        close("dbeta", db, beta.grad, rel=2e-4, abs_=1e-4)


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("shape,affine", [((1, 128, 16, 16, 16), False), ((2, 64, 5, 7, 6), False), ((1, 512, 8, 8, 8), True),
                                          ((1, 32, 1, 1, 3), False)])
def test_small_instance_norm_backward_in_one_launch(shape, affine, stored):
    """mmtta_norm_bwd_small (the deep levels: one workgroup per 32 channels, reduce + finalize + apply in one launch) against
    torch autograd of relu(instance_norm(y) * gamma + beta) and against the three-pass form (same formulas, another fp32
    summation order); fp32- and bf16-stored y; not eligible: > 4096 voxels, C no multiple of 32."""
    from multimodal_tta_amd import ops

    torch.manual_seed(9)
    n, c, d, h, w = shape
    y = (torch.randn(shape) * 1.7 + 0.3)
    if stored == "bf16":
        y = y.to(torch.bfloat16).float()
    y.requires_grad_(True)
    gamma = (torch.rand(c) + 0.5) if affine else None
    beta = (torch.randn(c) * 0.1) if affine else None
    ref = F.relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5))
    gout = torch.randn_like(ref)
    ref.backward(gout)
    y32 = cl(y.detach())
    if stored == "bf16":
        y_cl = ops.new_cl(n, d, h, w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), dtype=torch.bfloat16)
        y_cl.copy_(y32)
    else:
        y_cl = y32
    rows = ops.reduce_rows_per_n(y32)
    part = torch.empty(n * rows * 2 * c, device="cuda")
    ops.channel_stats(y32, part)
    mean, rstd = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    scratch = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ops.norm_stats_finalize(ops.NORM_INSTANCE, 1, part, rows, n, c, d * h * w, 1e-5, True, None, None, 0.1, mean, rstd, scratch)
    g_d = gamma.cuda() if affine else None
    b_d = beta.cuda() if affine else None
    nl = ops.NL(mean, rstd, g_d, b_d, relu=True)
    dT = cl(gout)
    dy1 = torch.empty_like(y32)
    assert ops.norm_bwd_small_ok(dT, y_cl, nl, dy1)
    ops.norm_bwd_small(dT, y_cl, nl, d * h * w, dy1)
    bpart = torch.empty(n * rows * 2 * c, device="cuda")
    m1, m2 = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    ops.norm_bwd_reduce(dT, y_cl, nl, bpart)
    ops.norm_bwd_finalize(ops.NORM_INSTANCE, 1, bpart, rows, n, c, d * h * w, g_d, True, m1, m2, None, None, False, scratch)
    dy3 = torch.empty_like(y32)
    ops.norm_bwd_apply(dT, y_cl, nl, m1, m2, dy3)
    torch.cuda.synchronize()
    close("one launch vs autograd", ncdhw(dy1), y.grad, rel=2e-4, abs_=2e-6)
    close("one launch vs three passes", dy1, dy3, rel=2e-5, abs_=1e-6)
    # in place (dy aliases dout), as the engine may call it
    ops.norm_bwd_small(dT, y_cl, nl, d * h * w, dT)
    torch.cuda.synchronize()
    assert torch.equal(dT, dy1)
    big = torch.empty(1, 17, 16, 16, 32, device="cuda")
    assert not ops.norm_bwd_small_ok(big, big, ops.NL(mean[:32], rstd[:32], None, None, relu=True), big)
    odd = torch.empty(1, 4, 4, 4, 40, device="cuda")
    assert not ops.norm_bwd_small_ok(odd, odd, ops.NL(mean[:40], rstd[:40], None, None, relu=True), odd)


def test_batchnorm_eval_uses_running_stats():
    from multimodal_tta_amd import ops
    torch.manual_seed(5)
    n, c, d, h, w = 2, 4, 4, 4, 4
    y = torch.randn(n, c, d, h, w)
    rm, rv = torch.randn(c) * 0.2, torch.rand(c) + 0.5
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.1
    ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training=False, eps=1e-5))
    mean, rstd = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    scratch = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ops.norm_stats_finalize(ops.NORM_BATCH, 1, None, 0, n, c, d * h * w, 1e-5, False, rm.cuda(), rv.cuda(), 0.1, mean,
                            rstd, scratch)
    y_cl = cl(y)
    out = torch.empty_like(y_cl)
    ops.combine(y_cl, ops.NL(mean, rstd, gamma.cuda(), beta.cuda(), True), None, None, out)
    torch.cuda.synchronize()
    close("bn eval", ncdhw(out), ref)


def test_combine_two_sources_and_slices():
    from multimodal_tta_amd import ops
    torch.manual_seed(9)
    n, c, d, h, w = 1, 6, 4, 5, 6     # c % 4 != 0 -> scalar path
    a, b = torch.randn(n, c, d, h, w), torch.randn(n, c, d, h, w)
    ref = F.relu(F.instance_norm(a)) + b
    mu = a.mean(dim=(2, 3, 4)).reshape(-1).cuda()
    rs = (1 / torch.sqrt(a.var(dim=(2, 3, 4), unbiased=False) + 1e-5)).reshape(-1).cuda()
    wide = torch.zeros(n, d, h, w, 16, device="cuda")
    ops.combine(cl(a), ops.NL(mu, rs, relu=True), cl(b), None, wide[..., 4:10])
    torch.cuda.synchronize()
    close("combine", ncdhw(wide[..., 4:10]), ref)
    assert wide[..., :4].abs().max().item() == 0 and wide[..., 10:].abs().max().item() == 0


def test_upsample_backward_from_a_channel_slice():
    """The adjoint reads its gradient out of a channel slice of the concat gradient (deep-fusion decoder)."""
    from multimodal_tta_amd import ops
    torch.manual_seed(12)
    x = torch.randn(1, 4, 16, 16, 16, requires_grad=True)
    up = torch.nn.Upsample(scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=True)
    ref = up(x)
    g = torch.randn_like(ref)
    ref.backward(g)
    wide = torch.randn(1, 32, 32, 32, 8, device="cuda")
    ops.to_cl(g.cuda(), out=wide[..., :4])
    dx = ops.new_cl(1, 16, 16, 16, 4, "cuda")
    ops.upsample2x_bwd(wide[..., :4], dx)
    torch.cuda.synchronize()
    close("upsample bwd from slice", ncdhw(dx), x.grad, rel=1e-5, abs_=1e-5)


@pytest.mark.parametrize("shape", [(1, 3, 3, 4, 5), (2, 8, 4, 4, 4), (1, 4, 1, 2, 3), (1, 4, 16, 16, 16), (1, 2, 8, 24, 24)])
def test_upsample_trilinear(shape):
    from multimodal_tta_amd import ops
    torch.manual_seed(2)
    x = torch.randn(shape, requires_grad=True)
    up = torch.nn.Upsample(scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=True)
    ref = up(x)
    g = torch.randn_like(ref)
    ref.backward(g)
    n, c, d, h, w = shape
    x_cl = cl(x.detach())
    y_cl = ops.new_cl(n, 2 * d, 2 * h, 2 * w, c, "cuda")
    ops.upsample2x_fwd(x_cl, y_cl)
    dx_cl = ops.new_cl(n, d, h, w, c, "cuda")
    ops.upsample2x_bwd(cl(g), dx_cl)
    torch.cuda.synchronize()
    close("upsample fwd", ncdhw(y_cl), ref, rel=1e-5, abs_=1e-6)
    close("upsample bwd", ncdhw(dx_cl), x.grad, rel=1e-5, abs_=1e-5)


@pytest.mark.parametrize("shape", [(1, 32, 4, 6, 8), (2, 33, 3, 4, 5), (1, 8, 8, 8, 8)])
def test_upsample_backward_with_bf16_stored_gradients(shape):
    """method.grad_storage: bf16 in the deep-fusion decoder - the adjoint of the trilinear resample reads a bf16-stored
    gradient (also a channel slice of the concat gradient: 33 channels in rows of 40) and writes a bf16-stored one, fp32
    arithmetic in between: equal to round_bf16(fp32 result on the same bf16 input) to one bf16 ulp; accumulate included."""
    from multimodal_tta_amd import ops
    torch.manual_seed(21)
    n, c, d, h, w = shape
    x = torch.randn(n, c, d, h, w, requires_grad=True)
    up = torch.nn.Upsample(scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=True)
    ref = up(x)
    g = torch.randn_like(ref).to(torch.bfloat16).float()
    ref.backward(g)
    g16 = ops.new_cl(n, 2 * d, 2 * h, 2 * w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), zero=True, dtype=torch.bfloat16)
    g16.copy_(g.permute(0, 2, 3, 4, 1).to(torch.bfloat16))
    dx = ops.new_cl(n, d, h, w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), zero=True, dtype=torch.bfloat16)
    ops.upsample2x_bwd(g16, dx)
    torch.cuda.synchronize()
    got = dx.float().permute(0, 4, 1, 2, 3).cpu()
    scale = x.grad.abs().max().item()
    assert (got - x.grad).abs().max().item() <= 2.0 ** -8 * scale, "bf16-stored adjoint"
    ops.upsample2x_bwd(g16, dx, accumulate=True)
    torch.cuda.synchronize()
    got2 = dx.float().permute(0, 4, 1, 2, 3).cpu()
    assert (got2 - 2 * x.grad).abs().max().item() <= 3 * 2.0 ** -8 * scale, "accumulate in the gradient's storage"
    with pytest.raises(Exception):
        ops.upsample2x_bwd(g16, ops.new_cl(n, d, h, w, c, "cuda"))          # mixed storage is refused, loudly


def _thin_bf16(t_ncdhw):
    """NCDHW cpu fp32 (<= 4 channels) -> bf16-stored channels-last cuda view with 8-byte voxels that owns its pad."""
    from multimodal_tta_amd import ops
    n, c, d, h, w = t_ncdhw.shape
    out = ops.new_cl(n, d, h, w, c, "cuda", ldc=4, zero=True, dtype=torch.bfloat16)
    ops.to_cl(t_ncdhw.cuda().contiguous(), out=out)
    return out


@pytest.mark.parametrize("shape", [(1, 3, 8, 8, 8), (2, 1, 5, 6, 7), (1, 4, 4, 9, 33)])
def test_thin_norm_backward_with_bf16_stored_gradients(shape):
    """Thin full-resolution tensors (<= 4 channels): the activation stays fp32-stored, the gradients going into and coming out
    of the norm backward are bf16-stored with 8-byte voxels.  With a bf16-representable incoming gradient the reduction sees
    the same values as the fp32-stored call (partials bitwise equal) and the result is round_bf16 of the fp32 result."""
    from multimodal_tta_amd import ops
    torch.manual_seed(33)
    n, c, d, h, w = shape
    y = torch.randn(shape) * 1.7 + 0.3
    gout = torch.randn(shape).to(torch.bfloat16).float()
    y32, g32, g16 = cl(y), cl(gout), _thin_bf16(gout)
    rows = ops.reduce_rows_per_n(y32)
    part = torch.empty(n * rows * 2 * c, device="cuda")
    ops.channel_stats(y32, part)
    mean, rstd = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    scratch = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ops.norm_stats_finalize(ops.NORM_INSTANCE, 1, part, rows, n, c, d * h * w, 1e-5, True, None, None, 0.1, mean, rstd, scratch)
    nl = ops.NL(mean, rstd, None, None, relu=True)
    out = {}
    for name, gg in (("fp32", g32), ("bf16", g16)):
        bpart = torch.empty(n * rows * 2 * c, device="cuda")
        m1, m2 = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
        ops.norm_bwd_reduce(gg, y32, nl, bpart)
        ops.norm_bwd_finalize(ops.NORM_INSTANCE, 1, bpart, rows, n, c, d * h * w, None, True, m1, m2, None, None, False, scratch)
        dy = ops.new_cl(n, d, h, w, c, "cuda", ldc=4, zero=True, dtype=gg.dtype)
        ops.norm_bwd_apply(gg, y32, nl, m1, m2, dy)
        torch.cuda.synchronize()
        out[name] = (bpart.clone(), dy.float().clone())
    assert torch.equal(out["fp32"][0], out["bf16"][0]), "reduction partials differ between the storages"
    assert torch.equal(out["bf16"][1], out["fp32"][1].to(torch.bfloat16).float()), "apply: not round_bf16 of the fp32 result"


@pytest.mark.parametrize("shape", [(1, 3, 8, 8, 8), (2, 1, 5, 6, 7), (1, 4, 4, 9, 33)])
def test_entropy_gradient_into_a_bf16_stored_tensor(shape):
    """The Bernoulli objective writing d(logits) bf16-stored (8-byte voxels): the same loss, round_bf16 of the fp32 gradient;
    per-item objectives included.  The categorical objective refuses a bf16 destination."""
    from multimodal_tta_amd import ops
    torch.manual_seed(8)
    n, c, d, h, w = shape
    z = torch.randn(shape) * 3.0
    z_cl = cl(z)
    res = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        g = ops.new_cl(n, d, h, w, c, "cuda", ldc=4, zero=True, dtype=dt)
        partial = torch.empty(ops.entropy_partials_items(z_cl), dtype=torch.float64, device="cuda")
        loss = torch.empty(n, device="cuda")
        ops.entropy_loss_items(z_cl, g, partial, loss, softmax=False)
        torch.cuda.synchronize()
        res[name] = (loss.clone(), g.float().clone())
    assert torch.equal(res["fp32"][0], res["bf16"][0])
    assert torch.equal(res["bf16"][1], res["fp32"][1].to(torch.bfloat16).float())
    with pytest.raises(Exception):
        g = ops.new_cl(n, d, h, w, c, "cuda", ldc=4, zero=True, dtype=torch.bfloat16)
        partial = torch.empty(ops.entropy_partials(z_cl), dtype=torch.float64, device="cuda")
        ops.entropy_loss(z_cl, g, partial, torch.empty(1, device="cuda"), softmax=True)


def test_lincomb_mean_and_accumulate():
    from multimodal_tta_amd import ops
    torch.manual_seed(4)
    xs = [torch.randn(1, 8, 4, 4, 4) for _ in range(4)]
    out = ops.new_cl(1, 4, 4, 4, 8, "cuda")
    ops.lincomb([cl(t) for t in xs], [0.25] * 4, out)
    torch.cuda.synchronize()
    close("mean of 4", ncdhw(out), torch.stack(xs).mean(0))
    ops.lincomb([cl(xs[0])], [2.0], out, accumulate=True)
    torch.cuda.synchronize()
    close("accumulate", ncdhw(out), torch.stack(xs).mean(0) + 2 * xs[0])


@pytest.mark.parametrize("softmax", [False, True])
@pytest.mark.parametrize("shape", [(1, 3, 8, 8, 8), (2, 1, 5, 6, 7), (1, 4, 16, 16, 16)])
def test_entropy_loss(softmax, shape):
    from multimodal_tta_amd import ops
    torch.manual_seed(6)
    z = (torch.randn(shape) * 3).requires_grad_(True)
    if softmax:
        logp = F.log_softmax(z, dim=1)
        ref = -(logp.exp() * logp).sum(1).mean()
    else:
        ref = (F.softplus(z) - z * torch.sigmoid(z)).mean()
    ref.backward()
    z_cl = cl(z.detach())
    g_cl = torch.empty_like(z_cl)
    partial = torch.empty(ops.entropy_partials(z_cl), dtype=torch.float64, device="cuda")
    loss = torch.empty(1, device="cuda")
    ops.entropy_loss(z_cl, g_cl, partial, loss, softmax=softmax)
    torch.cuda.synchronize()
    close("loss", loss.cpu().reshape(()), ref, rel=2e-6, abs_=1e-7)
    close("dloss/dlogits", ncdhw(g_cl), z.grad, rel=2e-5, abs_=1e-9)


def test_adam_matches_torch_over_steps():
    """Two-segment fused Adam == torch.optim.Adam with the reference's decay / no-decay groups
    (reference src/core/experiment_manager.py:214-228; defaults configs/training/default.yaml:30-39)."""
    from multimodal_tta_amd import ops
    torch.manual_seed(8)
    n_decay, n_nodecay = 1000, 37
    n = n_decay + n_nodecay
    p0 = torch.randn(n)
    pd = torch.nn.Parameter(p0[:n_decay].clone())
    pn = torch.nn.Parameter(p0[n_decay:].clone())
    lr, betas, eps, wd = 1e-3, (0.9, 0.9999), 1e-8, 5e-4
    opt = torch.optim.Adam([{"params": [pd], "weight_decay": wd}, {"params": [pn], "weight_decay": 0.0}], lr=lr,
                           betas=betas, eps=eps)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(5):
        g = torch.randn(n) * (10.0 ** (-t))
        pd.grad, pn.grad = g[:n_decay].clone(), g[n_decay:].clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, n_decay, lr, betas[0], betas[1], eps, wd, step)
    torch.cuda.synchronize()
    assert int(step.item()) == 5
    ref = torch.cat([pd.detach(), pn.detach()])
    close("adam params", p, ref, rel=1e-6, abs_=2e-7)


@pytest.mark.parametrize("name,kw", [
    ("adam", dict(lr=1e-3, betas=(0.9, 0.9999), eps=1e-8, weight_decay=5e-4)),
    ("adamw", dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)),
    ("sgd", dict(lr=1e-2, momentum=0.9, weight_decay=1e-4, dampening=0.0, nesterov=False)),
    ("sgd", dict(lr=1e-2, momentum=0.9, weight_decay=1e-4, dampening=0.0, nesterov=True)),
    ("sgd", dict(lr=1e-2, momentum=0.8, weight_decay=0.0, dampening=0.3, nesterov=False)),
    ("sgd", dict(lr=1e-2, momentum=0.0, weight_decay=1e-3, dampening=0.0, nesterov=False)),
])
def test_fused_optimizers_match_torch_over_steps(name, kw):
    """mmtta_optim_step == torch.optim.{Adam, AdamW, SGD} with the reference's decay / no-decay groups over 5 steps
    (the three classes of the reference factory, src/core/experiment_manager.py:199-210; hyper-parameters of
    configs/training/default.yaml:13-45)."""
    from multimodal_tta_amd import ops
    torch.manual_seed(8)
    n_decay, n_nodecay = 1000, 36        # arena segments are multiples of 4
    n = n_decay + n_nodecay
    p0 = torch.randn(n)
    pd = torch.nn.Parameter(p0[:n_decay].clone())
    pn = torch.nn.Parameter(p0[n_decay:].clone())
    cls = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "sgd": torch.optim.SGD}[name]
    tkw = dict(kw)
    wd = tkw.pop("weight_decay")
    opt = cls([{"params": [pd], "weight_decay": wd}, {"params": [pn], "weight_decay": 0.0}], **tkw)
    b = kw.get("betas", (0.9, 0.999))
    spec = ops.OptimSpec(name=name, lr=kw["lr"], beta1=b[0], beta2=b[1], eps=kw.get("eps", 1e-8), weight_decay=wd,
                         momentum=kw.get("momentum", 0.0), dampening=kw.get("dampening", 0.0),
                         nesterov=kw.get("nesterov", False))
    p = p0.clone().cuda()
    # the state buffers hold garbage: the call that finds step == 0 starts from zero moments and does not read them (the
    # episodic reset clears the step counter only)
    m, v = torch.full((n,), float("nan"), device="cuda"), torch.full((n,), float("inf"), device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(5):
        g = torch.randn(n) * (10.0 ** (-t))
        pd.grad, pn.grad = g[:n_decay].clone(), g[n_decay:].clone()
        opt.step()
        ops.optim_step(spec, p, g.cuda(), m, v, n_decay, step)
    torch.cuda.synchronize()
    assert int(step.item()) == 5
    close(f"{name} params", p, torch.cat([pd.detach(), pn.detach()]), rel=1e-6, abs_=2e-7)


@pytest.mark.parametrize("R,thr,shape", [(3, 0.5, (2, 8, 8, 8)), (1, 0.3, (1, 6, 10, 12)), (3, 0.5, (1, 32, 32, 32))])
def test_mask_dice_counts_exact(R, thr, shape):
    """Integer counts are bit exact against the reference formulas (src/evaluation/seg_eval.py:41-68,304-306)."""
    from multimodal_tta_amd import ops
    torch.manual_seed(10)
    n, d, h, w = shape
    z = torch.randn(n, R, d, h, w) * 2
    z[0, 0, 0, 0, :4] = torch.tensor([0.0, -1e-9, 1e-9, math.log(thr / (1 - thr))])
    lab = (torch.rand(n, R, d, h, w) > 0.6).float()
    lab[0, R - 1] = 0.0   # empty-GT region
    pred = (torch.sigmoid(z) >= thr).to(torch.uint8)
    gt = (lab > 0.5).to(torch.uint8)
    pf, gf = pred.reshape(n, R, -1).float(), gt.reshape(n, R, -1).float()
    ref = torch.stack([(pf * gf).sum(-1), pf.sum(-1), gf.sum(-1)], dim=-1).long()
    counts = torch.empty(n, R, 3, dtype=torch.int64, device="cuda")
    mask = torch.empty(n, R, d, h, w, dtype=torch.uint8, device="cuda")
    ops.mask_dice_counts(cl(z), lab.cuda(), thr, counts, mask)
    torch.cuda.synchronize()
    mism = (mask.cpu() != pred).sum().item()
    assert mism <= 1, f"{mism} mask voxels differ (only the exact-threshold voxel may, by 1 ulp of expf)"
    if mism == 0:
        assert torch.equal(counts.cpu(), ref)


def test_intensity_normalisation_matches_the_oracle():
    """Input pre-pass (reference src/datasets/transforms.py:129-223): clip + masked z-score per channel, the
    min_count fallback, clip only, and the legacy mean/std branch.  The statistics are fp64 sums on the GPU and a
    two-pass fp32 mean / std in torch: agreement to 2e-5 relative of the normalised range."""
    import oracle
    from multimodal_tta_amd.transforms import normalize_image
    torch.manual_seed(21)
    ct = torch.randn(1, 12, 36, 40) * 600 - 300
    ct[:, :3] = -1024.0                                       # air: below mask_gt
    pt = torch.rand(1, 12, 36, 40) * 20
    pt[:, :, :18] = 0.0                                       # background
    img = torch.cat([ct, pt], 0)
    pol = {"enabled": True, "channel_names": ["ct", "pt"],
           "channels": {"ct": {"clip": [-1000, 1000], "zscore": {"masked": True, "mask_gt": -900, "eps": 1.0e-6}},
                        "pt": {"clip": [0.0, 15.0], "zscore": {"masked": True, "mask_gt": 0.0, "eps": 1.0e-6}}}}
    close("hecktor policy", normalize_image(img.cuda(), intensity_policy=pol), oracle.normalize_image(img, intensity_policy=pol),
          rel=2e-5, abs_=2e-5)
    sparse = torch.zeros(2, 4, 8, 8)
    sparse[0, 0, 0, :5] = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0])     # 5 voxels > 0 < min_count 16: all-voxel statistics
    sparse[1] = 7.0                                                    # constant channel: sd -> eps
    pol2 = {"enabled": True, "channels": {"0": {"zscore": {"masked": True, "mask_gt": 0.0}},
                                          "1": {"clip": [0.0, 5.0]}}}
    close("fallbacks", normalize_image(sparse.cuda(), intensity_policy=pol2), oracle.normalize_image(sparse, intensity_policy=pol2),
          rel=2e-5, abs_=2e-5)
    x4 = torch.randn(4, 6, 10, 12)
    close("legacy", normalize_image(x4.cuda(), mean=[0.1, 0.2, 0.3, 0.4], std=[1.0, 2.0, 0.5, 4.0]),
          oracle.normalize_image(x4, mean=[0.1, 0.2, 0.3, 0.4], std=[1.0, 2.0, 0.5, 4.0]), rel=1e-6, abs_=1e-6)
    assert normalize_image(x4.cuda(), normalize=False).data_ptr() != 0


@pytest.mark.parametrize("shape", [(1, 64, 16, 16, 16), (2, 32, 9, 10, 11), (1, 512, 8, 8, 8), (1, 24, 6, 6, 7)])
def test_norm_backward_with_bf16_stored_gradients(shape):
    """method.grad_storage: bf16 - the gradient going INTO the norm backward and the one coming out are bf16-stored (next to a
    bf16-stored activation).  With a bf16-representable incoming gradient the three-pass form (reduce / finalize / apply:
    octet kernel, generic kernel for C = 24) and the one-launch form compute what the fp32-stored forms compute, the result
    rounded to bf16 once on the way out: equal to round_bf16(fp32 result) to one bf16 ulp (the reductions see the same
    values; only the apply's fp32 summation order inside m1 / m2 may differ in the last fp32 bit)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(21)
    n, c, d, h, w = shape
    y = (torch.randn(shape) * 1.7 + 0.3).to(torch.bfloat16).float()
    gout = torch.randn(shape).to(torch.bfloat16).float()
    y32, g32 = cl(y), cl(gout)
    y16 = ops.new_cl(n, d, h, w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), dtype=torch.bfloat16)
    y16.copy_(y32)
    g16 = ops.new_cl(n, d, h, w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), dtype=torch.bfloat16)
    g16.copy_(g32)
    rows = ops.reduce_rows_per_n(y32)
    part = torch.empty(n * rows * 2 * c, device="cuda")
    ops.channel_stats(y32, part)
    mean, rstd = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
    scratch = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ops.norm_stats_finalize(ops.NORM_INSTANCE, 1, part, rows, n, c, d * h * w, 1e-5, True, None, None, 0.1, mean, rstd, scratch)
    nl = ops.NL(mean, rstd, None, None, relu=True)
    out = {}
    for name, yy, gg in (("fp32", y16, g32), ("bf16", y16, g16)):
        bpart = torch.empty(n * rows * 2 * c, device="cuda")
        m1, m2 = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
        ops.norm_bwd_reduce(gg, yy, nl, bpart)
        ops.norm_bwd_finalize(ops.NORM_INSTANCE, 1, bpart, rows, n, c, d * h * w, None, True, m1, m2, None, None, False, scratch)
        dy = torch.empty_like(gg)
        ops.norm_bwd_apply(gg, yy, nl, m1, m2, dy)
        torch.cuda.synchronize()
        out[name] = (bpart.clone(), dy.float().clone())
        if c % 32 == 0 and d * h * w <= 4096:
            dy1 = torch.empty_like(gg)
            assert ops.norm_bwd_small_ok(gg, yy, nl, dy1)
            ops.norm_bwd_small(gg, yy, nl, d * h * w, dy1)
            torch.cuda.synchronize()
            out[name + "_small"] = dy1.float().clone()
    assert torch.equal(out["fp32"][0], out["bf16"][0]), "the reduction sees the same gradient values in either storage"
    want = out["fp32"][1].to(torch.bfloat16).float()
    ulp = want.abs() * 2.0 ** -7 + 1e-30
    assert bool(((out["bf16"][1] - want).abs() <= ulp).all()), "bf16-stored result differs from round_bf16(fp32 result)"
    if "bf16_small" in out:
        want1 = out["fp32_small"].to(torch.bfloat16).float()
        assert bool(((out["bf16_small"] - want1).abs() <= want1.abs() * 2.0 ** -7 + 1e-30).all())
    # a bf16 gradient next to an fp32-stored activation is refused (the engine never builds that pair)
    with pytest.raises(ops.MmttaError):
        ops.norm_bwd_reduce(g16, y32, nl, part)
