"""Per-volume parameter sets (``mmtta_param_sets``, ``method.group``): N volumes - or the M identical modality encoders
of the deep-fusion network - as the batch items of ONE launch, each item with its own weights.

The contract (include/mmtta.h): every batch item is computed exactly as if it had been launched alone with its
parameter set - same tiles, same reduction splits, same summation order.  So the checks here are BITWISE: a grouped
launch against one launch per item, for every kernel family the convolution entry points dispatch to, for the weight
gradient (one set per item, and sets shared by consecutive items), the per-item entropy objective, the replicated
optimizer, and end to end (a group of volumes through the adaptation plugin against the same volumes one at a time).
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_hip_conv import cl, ref_module  # noqa: E402

# (cin, cout, k, stride, transposed, (d, h, w)): one case per kernel family behind mmtta_conv_run / mmtta_conv_wgrad
GROUP_CASES = [
    (32, 32, 3, 1, False, (8, 8, 16)),       # implicit GEMM, lean / 8x8x8 tile; transposed-read weight gradient
    (64, 64, 3, 1, False, (8, 8, 8)),        # two column blocks
    (128, 136, 3, 1, False, (4, 6, 8)),      # wide, ragged columns, split-K + finalize
    (512, 512, 3, 1, False, (4, 4, 4)),      # the 8^3-level shape class: split-K
    (32, 64, 3, 2, False, (8, 8, 16)),       # stride 2; its input gradient = 8 parity classes
    (64, 32, 3, 2, True, (4, 4, 8)),         # ConvTranspose3d
    (256, 512, 1, 1, False, (4, 4, 4)),      # 1x1x1
    (4, 32, 3, 2, False, (16, 16, 16)),      # thin-K first layer (lanes along N)
    (1, 32, 3, 2, False, (8, 8, 8)),         # one-channel stem of a modality encoder
    (64, 3, 3, 2, True, (3, 5, 70)),         # full-resolution up-convolution to <= 4 channels
    (3, 3, 3, 1, False, (8, 8, 8)),          # R -> R row / matrix-tile kernels, tiny weight gradient
    (32, 3, 1, 1, False, (4, 6, 50)),        # 1x1 head (lanes along K)
    (5, 4, 3, 1, False, (8, 8, 8)),          # generic direct kernel
    (33, 32, 3, 1, False, (8, 8, 16)),       # ragged reduction (decoder of the deep-fusion net)
]


def _sets(op, items_per_set, inner, w_outer, w_inner, b_outer, b_inner, on=True):
    class Ctl:
        use_sets = on
    op.set_param_sets(items_per_set, inner, w_outer, w_inner, b_outer, b_inner, Ctl())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", GROUP_CASES)
def test_grouped_conv_equals_one_launch_per_item(cin, cout, k, stride, transposed, shape, dtype):
    """forward (+ statistics rows, fused add), input gradient, weight and bias gradient: G = 3 items with 3 parameter sets
    in one launch == 3 launches of one item each, bit for bit."""
    from multimodal_tta_amd import ops

    G = 3
    d, h, w = shape
    torch.manual_seed(5 + cin + 3 * cout)
    dt = ops.PRECISIONS[dtype]
    mods = [ref_module(cin, cout, k, stride, transposed) for _ in range(G)]
    wshape = tuple(mods[0].weight.shape)
    wnum = mods[0].weight.numel()
    wpad = (wnum + 3) // 4 * 4
    bpad = (cout + 3) // 4 * 4
    W = torch.zeros(G, wpad, device="cuda")
    Bv = torch.zeros(G, bpad, device="cuda")
    for g, m in enumerate(mods):
        W[g, :wnum] = m.weight.detach().reshape(-1).cuda()
        Bv[g, :cout] = m.bias.detach().cuda()
    x = torch.randn(G, cin, d, h, w)
    x_cl = cl(x)

    def run(items):
        """items: list of batch indices launched together -> (y, stats, dx, dw, db) per item"""
        n = len(items)
        op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=dt, n_sets=n)
        for j, g in enumerate(items):
            op.pack(W[g, :wnum].view(wshape), j)
        # the bias vectors of this launch in their own compact buffer (stride bpad); dw / db come back at strides wpad / bpad
        Bl = torch.stack([Bv[g] for g in items]).contiguous()
        _sets(op, 1, 1, wpad, 0, bpad, 0, on=n > 1)
        xin = x_cl[items[0]:items[0] + 1] if n == 1 else x_cl
        n_, do, ho, wo, _ = op.out_shape(xin)
        y = ops.new_cl(n_, do, ho, wo, cout, "cuda")
        rows = op.stats_rows(xin, y)
        stats = torch.zeros((rows, 2, cout), device="cuda")
        op.forward(xin, None, Bl[0, :cout], y, stats=stats)
        gen = torch.Generator().manual_seed(99)
        gy_all = torch.randn(G, cout, do, ho, wo, generator=gen)
        gy = cl(gy_all if n > 1 else gy_all[items[0]:items[0] + 1])
        dx = ops.new_cl(n_, d, h, w, cin, "cuda")
        op.dgrad(gy, dx)
        dw = torch.full((n, wpad), float("nan"), device="cuda")
        db = torch.full((n, bpad), float("nan"), device="cuda")
        op.wgrad(xin, None, gy, dw[0, :wnum].view(wshape), db[0, :cout])
        torch.cuda.synchronize()
        return y.clone(), stats.view(n_, rows // n_, 2, cout).clone(), dx.clone(), dw[:, :wnum].clone(), db[:, :cout].clone()

    together = run(list(range(G)))
    for g in range(G):
        alone = run([g])
        for name, a, b in zip(("forward", "statistics rows", "input gradient", "weight gradient", "bias gradient"), alone, together):
            assert torch.equal(a[0], b[g]), f"{name} of item {g} differs between the grouped launch and the item alone"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_family_sets_and_shared_sets(dtype):
    """Two-level sets (G volumes x M family members: the modality encoders) and sets shared by M consecutive items (the
    fusion layer, applied M times with one weight per volume: its weight gradient sums over the M items of a volume)."""
    from multimodal_tta_amd import ops

    G, M, cin, cout, k = 2, 3, 32, 64, 3
    d = h = w = 8
    dt = ops.PRECISIONS[dtype]
    torch.manual_seed(11)
    mods = [[ref_module(cin, cout, k, 1, False) for _ in range(M)] for _ in range(G)]
    wshape, wnum = tuple(mods[0][0].weight.shape), mods[0][0].weight.numel()
    # arena-like layout: replica stride RS, members at stride wnum (weights) / cout (biases) inside a replica
    RS = M * wnum + M * cout + 64
    arena = torch.zeros(G, RS, device="cuda")
    grads = torch.full((G, RS), float("nan"), device="cuda")
    boff = M * wnum
    for g in range(G):
        for m in range(M):
            arena[g, m * wnum:(m + 1) * wnum] = mods[g][m].weight.detach().reshape(-1).cuda()
            arena[g, boff + m * cout:boff + (m + 1) * cout] = mods[g][m].bias.detach().cuda()
    x_cl = cl(torch.randn(G * M, cin, d, h, w))
    gy = cl(torch.randn(G * M, cout, d, h, w))

    # (a) family: item n = g*M + m uses set (g, m)
    op = ops.ConvOp(cin, cout, k, 1, False, "cuda", dtype=dt, n_sets=G * M)
    for g in range(G):
        for m in range(M):
            op.pack(arena[g, m * wnum:(m + 1) * wnum].view(wshape), g * M + m)
    _sets(op, 1, M, RS, wnum, RS, cout)
    y = ops.new_cl(G * M, d, h, w, cout, "cuda")
    op.forward(x_cl, None, arena[0, boff:boff + cout], y)
    op.wgrad(x_cl, None, gy, grads[0, :wnum].view(wshape), grads[0, boff:boff + cout])
    torch.cuda.synchronize()
    for g in range(G):
        for m in range(M):
            n = g * M + m
            o1 = ops.ConvOp(cin, cout, k, 1, False, "cuda", dtype=dt)
            o1.pack(arena[g, m * wnum:(m + 1) * wnum].view(wshape))
            y1 = ops.new_cl(1, d, h, w, cout, "cuda")
            o1.forward(x_cl[n:n + 1], None, arena[g, boff + m * cout:boff + (m + 1) * cout], y1)
            dw1 = torch.empty(wshape, device="cuda")
            db1 = torch.empty(cout, device="cuda")
            o1.wgrad(x_cl[n:n + 1], None, gy[n:n + 1], dw1, db1)
            torch.cuda.synchronize()
            assert torch.equal(y1[0], y[n]), f"family forward (volume {g}, member {m})"
            assert torch.equal(dw1.reshape(-1), grads[g, m * wnum:(m + 1) * wnum]), f"family weight gradient ({g}, {m})"
            assert torch.equal(db1, grads[g, boff + m * cout:boff + (m + 1) * cout]), f"family bias gradient ({g}, {m})"

    # (b) shared: items [g*M, (g+1)*M) use set g; dw[g] = sum over its M items == the plain call on that batch of M
    op2 = ops.ConvOp(cin, cout, k, 1, False, "cuda", dtype=dt, n_sets=G)
    for g in range(G):
        op2.pack(arena[g, :wnum].view(wshape), g)
    _sets(op2, M, 1, RS, 0, RS, 0)
    y2 = ops.new_cl(G * M, d, h, w, cout, "cuda")
    op2.forward(x_cl, None, arena[0, boff:boff + cout], y2)
    grads.fill_(float("nan"))
    op2.wgrad(x_cl, None, gy, grads[0, :wnum].view(wshape), grads[0, boff:boff + cout])
    torch.cuda.synchronize()
    for g in range(G):
        o1 = ops.ConvOp(cin, cout, k, 1, False, "cuda", dtype=dt)
        o1.pack(arena[g, :wnum].view(wshape))
        sl = slice(g * M, (g + 1) * M)
        y1 = ops.new_cl(M, d, h, w, cout, "cuda")
        o1.forward(x_cl[sl], None, arena[g, boff:boff + cout], y1)
        dw1 = torch.empty(wshape, device="cuda")
        db1 = torch.empty(cout, device="cuda")
        o1.wgrad(x_cl[sl], None, gy[sl], dw1, db1)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2[sl]), f"shared-set forward (volume {g})"
        assert torch.equal(dw1.reshape(-1), grads[g, :wnum]), f"shared-set weight gradient (volume {g})"
        assert torch.equal(db1, grads[g, boff:boff + cout]), f"shared-set bias gradient (volume {g})"


@pytest.mark.parametrize("softmax", [False, True])
def test_entropy_per_item_equals_separate_calls(softmax):
    from multimodal_tta_amd import ops

    torch.manual_seed(3)
    N, R = 3, 3
    z = ops.new_cl(N, 12, 10, 18, R, "cuda", ldc=4, zero=True)
    z.copy_(torch.randn(N, 12, 10, 18, R, device="cuda") * 3)
    dz = ops.new_cl(N, 12, 10, 18, R, "cuda", ldc=4, zero=True)
    part = torch.empty(ops.entropy_partials_items(z), dtype=torch.float64, device="cuda")
    loss = torch.empty(N, device="cuda")
    ops.entropy_loss_items(z, dz, part, loss, softmax=softmax)
    for n in range(N):
        dz1 = ops.new_cl(1, 12, 10, 18, R, "cuda", ldc=4, zero=True)
        z1 = ops.new_cl(1, 12, 10, 18, R, "cuda", ldc=4, zero=True)
        z1.copy_(z[n:n + 1])
        p1 = torch.empty(ops.entropy_partials(z1), dtype=torch.float64, device="cuda")
        l1 = torch.empty(1, device="cuda")
        ops.entropy_loss(z1, dz1, p1, l1, softmax=softmax)
        torch.cuda.synchronize()
        assert torch.equal(l1[0], loss[n]) and torch.equal(dz1[0], dz[n]), f"item {n}"


@pytest.mark.parametrize("opt", ["adam", "adamw", "sgd"])
def test_optimizer_over_replicas_equals_one_call_per_replica(opt):
    from multimodal_tta_amd import ops

    torch.manual_seed(4)
    G, total, n, n_decay = 3, 1000, 900, 512
    spec = ops.OptimSpec(name=opt, lr=1e-2, weight_decay=1e-2, momentum=0.9 if opt == "sgd" else 0.0)
    p = torch.randn(G, total, device="cuda")
    g = torch.randn(G, total, device="cuda")
    m = torch.zeros(G, total, device="cuda")
    v = torch.zeros(G, total, device="cuda")
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    step, step1 = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(G, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ops.optim_step_sets(spec, p, g, m, v, n, n_decay, 2, step)          # only the first two replicas are in use
        for r in range(2):
            ops.optim_step(spec, p1[r, :n], g[r, :n], m1[r, :n], v1[r, :n], n_decay, step1[r:r + 1])
    torch.cuda.synchronize()
    assert int(step) == 3
    assert torch.equal(p, p1) and torch.equal(m, m1) and torch.equal(v, v1)


SMALL = dict(name="unet", in_channels=4, num_classes=3, spatial_dims=3, channels=[4, 8, 16, 32, 64],
             strides=[2, 2, 2, 2], num_res_units=2, norm="INSTANCE", act="RELU", dropout=0.0)
WIDE = dict(SMALL, channels=[32, 64, 128, 256, 512])


@pytest.mark.parametrize("model_cfg,shape,precision", [(SMALL, (32, 32, 32), "fp32"), (SMALL, (32, 32, 32), "bf16"),
                                                       (WIDE, (32, 32, 32), "bf16")])
def test_group_of_volumes_equals_one_volume_at_a_time(model_cfg, shape, precision):
    """`method.group: 3`: three different volumes adapt side by side (own replica of the weights and of the optimizer state)
    == the same three volumes through a `group: 1` plugin one after another - losses of every step and final logits bit
    for bit, graph replay included; a partial group (2 of 3) as well."""
    from multimodal_tta_amd.registry import get_plugin
    from test_hip_tta import build_pair, root_cfg, volume

    G = 3
    xs = [volume(i, shape)[0] for i in range(G)]
    outs = {}
    for group in (1, G):
        cfg = root_cfg(model_cfg, steps=3, lr=1e-3, precision=precision, group=group, tune_volumes=4)      # one launch geometry
        _, hip = build_pair(model_cfg)
        plug = get_plugin("entmin_tta")(cfg).setup(hip, "cuda")
        if group == 1:
            outs[1] = []
            for x in xs:
                r = plug.adapt_volume(x.cuda())
                outs[1].append((r["losses"].clone(), plug.logits(r).clone()))
        else:
            for rep in range(2):              # second pass: graph replay, episodic reset of every replica
                r = plug.adapt_volume(torch.cat(xs).cuda())
                outs[G] = (r["losses"].clone(), plug.logits(r).clone())
            r2 = plug.adapt_volume(torch.cat(xs[:2]).cuda())
            part = (r2["losses"].clone(), plug.logits(r2).clone())
        torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(outs[1][g][0], outs[G][0][:, g]), f"losses of volume {g}"
        assert torch.equal(outs[1][g][1][0], outs[G][1][g]), f"logits of volume {g}"
    for g in range(2):
        assert torch.equal(outs[1][g][0], part[0][:, g]) and torch.equal(outs[1][g][1][0], part[1][g]), f"partial group, volume {g}"


def test_models_with_norm_parameters_fall_back_to_one_volume_per_launch():
    """BatchNorm statistics run across the batch and norm affines are per-model parameters: such a model cannot adapt a group
    of volumes in one launch sequence - the plugin says so and falls back to method.group = 1 (lanes still apply)."""
    from multimodal_tta_amd.registry import get_plugin
    from test_hip_tta import build_pair, root_cfg

    cfg_m = dict(SMALL, norm="BATCH")
    cfg = root_cfg(cfg_m, steps=1, group=2)
    _, hip = build_pair(cfg_m)
    with pytest.warns(UserWarning, match="method.group = 2 -> 1"):
        plug = get_plugin("entmin_tta")(cfg).setup(hip, "cuda")
    assert plug.group == 1 and plug.rt.group == 1 and plug.rt.arena.replicas == 1
