"""Parity of the implicit-GEMM conv kernels (forward, input gradient, weight gradient) against
torch CPU fp32 (F.conv3d / F.conv_transpose3d + autograd), through the C ABI.

Tolerance: fp32 MFMA accumulates in k order like an fmaf chain; torch CPU (oneDNN) uses a
different summation order, so agreement is to rounding: |err| <= 2e-4 * max|ref| + 1e-5.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_REL, TOL_ABS = 2e-4, 1e-5


def close(name, got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= TOL_REL * scale + TOL_ABS, f"{name}: max|err|={err:.3e} (max|ref|={scale:.3e})"


def cl(x):  # NCDHW cpu -> channels-last cuda view
    from multimodal_tta_amd import ops
    return ops.to_cl(x.cuda().contiguous())


def ncdhw(x_cl):
    return x_cl.permute(0, 4, 1, 2, 3).contiguous().cpu()


CASES = [
    # (cin, cout, k, stride, transposed, (n, d, h, w))
    (4, 8, 3, 1, False, (1, 12, 10, 8)),
    (32, 32, 3, 1, False, (1, 16, 16, 16)),
    (33, 32, 3, 1, False, (1, 8, 8, 16)),
    (64, 64, 3, 1, False, (1, 8, 8, 8)),
    (128, 136, 3, 1, False, (2, 4, 6, 8)),
    (3, 3, 3, 1, False, (1, 8, 8, 8)),
    (4, 32, 3, 2, False, (1, 16, 16, 16)),
    (1, 32, 3, 2, False, (1, 8, 8, 8)),
    (32, 64, 3, 2, False, (1, 8, 8, 16)),
    (64, 128, 3, 2, False, (1, 5, 6, 7)),
    (256, 512, 1, 1, False, (1, 4, 4, 4)),
    (32, 3, 1, 1, False, (1, 8, 8, 8)),
    (512, 512, 3, 1, False, (1, 4, 4, 4)),
    (8, 3, 3, 2, True, (1, 5, 6, 4)),
    (64, 32, 3, 2, True, (1, 4, 4, 8)),
    (768, 128, 3, 2, True, (1, 2, 2, 2)),
    # the narrow layers of small deep-fusion decoders (1x1 pre-convs, 5-channel concat inputs)
    (8, 4, 1, 1, False, (1, 16, 16, 16)),
    (5, 4, 3, 1, False, (1, 16, 16, 16)),
    (5, 4, 1, 1, False, (1, 16, 16, 16)),
    (12, 8, 3, 1, False, (1, 8, 8, 8)),
    (16, 8, 1, 1, False, (1, 8, 8, 8)),
    (1, 4, 3, 2, False, (1, 16, 16, 16)),
    (4, 8, 3, 2, False, (1, 8, 8, 8)),
    # wide-K layers that produce <= 4 channels (lanes-along-K direct kernel): full-resolution up-convolution,
    # 1x1 heads; rows longer than one 64-voxel chunk and odd extents
    (64, 3, 3, 2, True, (1, 3, 5, 70)),
    (64, 1, 3, 2, True, (2, 4, 4, 8)),
    (32, 3, 1, 1, False, (1, 4, 6, 150)),
    (64, 2, 3, 1, False, (1, 5, 6, 7)),
    (3, 64, 3, 1, False, (1, 6, 6, 9)),
    # <= 4 channels on both sides (VALU weight gradient, row kernel forward / input gradient), rows > 64 voxels
    (4, 2, 3, 1, False, (2, 5, 6, 70)),
    (1, 1, 3, 1, False, (1, 6, 5, 9)),
    # the instantiations only the full-size network dispatches: igemm <4,4,4,4,8,32> (config 2 / 9),
    # weight-gradient launches with more than 32 slabs (slab_prereduce_kernel), stride-2 thin stage <1,1,4,4,8,*>
    (128, 128, 3, 1, False, (1, 32, 32, 32)),
    (768, 128, 3, 2, True, (1, 8, 8, 8)),
    (32, 32, 3, 1, False, (1, 32, 32, 32)),
    (16, 32, 3, 2, False, (1, 8, 8, 8)),
]


def ref_module(cin, cout, k, stride, transposed):
    pad = (k - 1) // 2
    if transposed:
        return torch.nn.ConvTranspose3d(cin, cout, k, stride=stride, padding=pad, output_padding=stride - 1)
    return torch.nn.Conv3d(cin, cout, k, stride=stride, padding=pad)


@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", CASES)
def test_conv_fwd_dgrad_wgrad(cin, cout, k, stride, transposed, shape):
    from multimodal_tta_amd import ops

    torch.manual_seed(1234 + cin * 7 + cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = torch.randn(n, cin, d, h, w, requires_grad=True)
    y_ref = mod(x)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)

    op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda")
    wt = mod.weight.detach().cuda().contiguous()
    bias = mod.bias.detach().cuda().contiguous()
    op.pack(wt)
    x_cl = cl(x.detach())
    y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda")
    rows = op.stats_rows(x_cl, y_cl)
    stats = torch.full((rows, 2, cout), float("nan"), device="cuda")
    op.forward(x_cl, None, bias, y_cl, stats=stats)
    torch.cuda.synchronize()
    close("forward", ncdhw(y_cl), y_ref)
    # per-tile statistics add up to the per-(n,c) sums of the output
    st = stats.view(n, rows // n, 2, cout).double().sum(1).cpu()
    yr = y_ref.detach().double()
    ref_sum = yr.sum(dim=(2, 3, 4))
    ref_sq = (yr * yr).sum(dim=(2, 3, 4))
    assert torch.allclose(st[:, 0], ref_sum, rtol=1e-3, atol=1e-2 * max(1.0, ref_sq.max().item()) ** 0.5), "stats sum"
    assert torch.allclose(st[:, 1], ref_sq, rtol=1e-3, atol=1e-3), "stats sumsq"

    gy_cl = cl(gy)
    dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
    op.dgrad(gy_cl, dx_cl)
    torch.cuda.synchronize()
    close("dgrad", ncdhw(dx_cl), x.grad)

    dw = torch.empty_like(wt)
    db = torch.empty_like(bias)
    op.wgrad(x_cl, None, gy_cl, dw, db)
    torch.cuda.synchronize()
    close("wgrad", dw, mod.weight.grad)
    close("bgrad", db, mod.bias.grad)
    # accumulate paths
    op.dgrad(gy_cl, dx_cl, accumulate=True)
    op.wgrad(x_cl, None, gy_cl, dw, db, accumulate=True)
    torch.cuda.synchronize()
    close("dgrad accumulate", ncdhw(dx_cl), 2 * x.grad)
    close("wgrad accumulate", dw, 2 * mod.weight.grad)


def test_conv_norm_on_load_and_epilogue():
    """y = conv(relu(instance_norm(x))) + relu(instance_norm(r)) with both transforms applied on load."""
    from multimodal_tta_amd import ops

    torch.manual_seed(7)
    n, cin, cout, d, h, w = 2, 32, 64, 6, 8, 8
    x = torch.randn(n, cin, d, h, w) * 2 + 0.5
    r = torch.randn(n, cout, d, h, w)
    mod = torch.nn.Conv3d(cin, cout, 3, padding=1)
    xin = F.relu(F.instance_norm(x))
    y_ref = mod(xin) + F.relu(F.instance_norm(r))

    def stats_of(t):
        mu = t.mean(dim=(2, 3, 4))
        var = t.var(dim=(2, 3, 4), unbiased=False)
        return mu.reshape(-1).cuda().contiguous(), (1.0 / torch.sqrt(var + 1e-5)).reshape(-1).cuda().contiguous()

    mx, rx = stats_of(x)
    mr, rr = stats_of(r)
    op = ops.ConvOp(cin, cout, 3, 1, False, "cuda")
    op.pack(mod.weight.detach().cuda().contiguous())
    x_cl, r_cl = cl(x), cl(r)
    y_cl = ops.new_cl(n, d, h, w, cout, "cuda")
    op.forward(x_cl, ops.NL(mx, rx, relu=True), mod.bias.detach().cuda(), y_cl, add=r_cl, add_nl=ops.NL(mr, rr, relu=True))
    torch.cuda.synchronize()
    close("fused forward", ncdhw(y_cl), y_ref)

    # weight gradient with the same norm-on-load of the input
    xin2 = xin.clone().requires_grad_(False)
    gy = torch.randn_like(y_ref)
    w_ = mod.weight.detach().clone().requires_grad_(True)
    F.conv3d(xin2, w_, None, padding=1).backward(gy)
    dw = torch.empty_like(w_, device="cuda")
    op.wgrad(x_cl, ops.NL(mx, rx, relu=True), cl(gy), dw, None)
    torch.cuda.synchronize()
    close("wgrad with norm on load", dw, w_.grad)


def test_conv_channel_slices():
    """Reading from and writing into channel slices of wider buffers (the concat replacement)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(11)
    n, d, h, w = 1, 8, 8, 8
    a = torch.randn(n, 32, d, h, w)
    b = torch.randn(n, 32, d, h, w)
    mod = torch.nn.Conv3d(64, 32, 3, padding=1)
    y_ref = mod(torch.cat([a, b], dim=1))
    cat = torch.empty(n, d, h, w, 64, device="cuda")
    ops.to_cl(a.cuda(), out=cat[..., :32])
    ops.to_cl(b.cuda(), out=cat[..., 32:])
    out = torch.zeros(n, d, h, w, 96, device="cuda")
    op = ops.ConvOp(64, 32, 3, 1, False, "cuda")
    op.pack(mod.weight.detach().cuda().contiguous())
    op.forward(cat, None, mod.bias.detach().cuda(), out[..., 32:64])
    torch.cuda.synchronize()
    close("slice forward", ncdhw(out[..., 32:64]), y_ref)
    assert out[..., :32].abs().max().item() == 0 and out[..., 64:].abs().max().item() == 0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_one_channel_slices_of_the_input_take_the_thin_kernel(precision):
    """The single-modality stems of the deep-fusion net (Conv3d 1->32 k3 s2 on channel m of the 4-channel input): a slice
    that starts inside a 16-byte group (m = 1..3) runs on the same thin-K kernel as m = 0 (plan config 13; it fell back
    to the padded fp32 implicit GEMM before) and matches torch."""
    from multimodal_tta_amd import ops

    torch.manual_seed(23)
    n, d, h, w = 1, 12, 10, 16
    x = torch.randn(n, 4, d, h, w)
    x_cl = cl(x)
    for m in range(4):
        mod = torch.nn.Conv3d(1, 32, 3, stride=2, padding=1)
        y_ref = mod(x[:, m:m + 1])
        op = ops.ConvOp(1, 32, 3, 2, False, "cuda", dtype=ops.BF16 if precision == "bf16" else ops.F32)
        op.pack(mod.weight.detach().cuda().contiguous())
        xs = x_cl[..., m:m + 1]
        y = torch.empty(n, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2, 32, device="cuda")
        assert int(op.plan(op.d_fwd, xs, y).config) == 13, f"modality {m}: not the thin-K kernel"
        op.forward(xs, None, mod.bias.detach().cuda(), y)
        torch.cuda.synchronize()
        close(f"modality {m} stem", ncdhw(y), y_ref)


BF16_CASES = [
    (32, 32, 3, 1, False, (1, 16, 16, 16)),
    (33, 32, 3, 1, False, (1, 8, 8, 16)),        # 16-channel tail step
    (64, 64, 3, 1, False, (1, 8, 8, 8)),
    (128, 136, 3, 1, False, (2, 4, 6, 8)),
    (32, 64, 3, 2, False, (1, 8, 8, 16)),
    (64, 128, 3, 2, False, (1, 5, 6, 7)),
    (512, 512, 3, 1, False, (1, 4, 4, 4)),       # split-K
    (256, 512, 1, 1, False, (1, 4, 4, 4)),
    (64, 32, 3, 2, True, (1, 4, 4, 8)),
    (768, 128, 3, 2, True, (1, 2, 2, 2)),
    (4, 32, 3, 2, False, (1, 16, 16, 16)),       # K <= 4: taps folded into the MFMA reduction (chan_mfma_kernel)
    (3, 32, 3, 1, False, (1, 5, 7, 11)),         # ragged tile borders, stride 1
    (1, 32, 3, 2, False, (2, 6, 10, 12)),        # single-modality encoder stem of the deep-fusion net
    (64, 3, 3, 2, True, (1, 5, 6, 70)),          # full-resolution up-convolution: bf16-staged activations, fp32 weights
    (32, 2, 3, 2, True, (2, 4, 4, 16)),
    (64, 1, 3, 2, True, (1, 3, 4, 9)),
    # full-size-only instantiations (see CASES)
    (128, 128, 3, 1, False, (1, 32, 32, 32)),
    (768, 128, 3, 2, True, (1, 8, 8, 8)),
    (32, 32, 3, 1, False, (1, 32, 32, 32)),
    (16, 32, 3, 2, False, (1, 8, 8, 8)),
    (256, 256, 3, 1, False, (1, 16, 16, 16)),
    (64, 3, 3, 2, True, (1, 3, 5, 70)),
    # <= 4 channels on both sides: v_mfma_f32_4x4x4_16B_bf16 (conv3_mfma4_kernel), ragged rows longer than one 64-voxel run
    (3, 3, 3, 1, False, (1, 5, 7, 70)),
    (4, 2, 3, 1, False, (2, 5, 6, 70)),
    (1, 1, 3, 1, False, (1, 6, 5, 9)),
    (2, 4, 3, 1, False, (1, 3, 9, 130)),
    (3, 3, 3, 1, False, (1, 8, 8, 8)),           # (CASES' shape: its weight gradient takes the transposed-read kernel in bf16 precision)
]


def cl_bf16(x):
    """NCDHW cpu fp32 -> bf16-stored channels-last cuda view (rows padded to 8 channels)."""
    from multimodal_tta_amd import ops
    n, c, d, h, w = x.shape
    out = ops.new_cl(n, d, h, w, c, "cuda", ldc=ops.row_pad(c, torch.bfloat16), zero=True, dtype=torch.bfloat16)
    out.copy_(x.permute(0, 2, 3, 4, 1).to(torch.bfloat16))
    return out


BF16_STORED = [c for c in BF16_CASES if c[0] >= 16 and c[1] > 4]      # layers whose input and output are "wide" activations


@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", BF16_STORED)
def test_conv_bf16_stored_activations(cin, cout, k, stride, transposed, shape):
    """`bf16` precision with bf16 STORAGE of the forward activations (x read as bf16, y and the fused residual operand
    written / read as bf16, statistics from the fp32 accumulators): forward, and the weight gradient reading the bf16 x,
    against torch fp32 on the bf16-rounded input.  Bound: 1.5e-2 * max|ref| like the bf16-operand test (the output is
    rounded once more when it is stored: 2^-9 relative)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(77 + cin + cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = torch.randn(n, cin, d, h, w).to(torch.bfloat16).float()          # exactly representable: isolates the kernel
    x.requires_grad_(True)
    res = torch.randn_like(mod(x.detach())).to(torch.bfloat16).float()
    y_ref = mod(x) + res
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
    wt = mod.weight.detach().cuda().contiguous()
    op.pack(wt)
    x_cl, r_cl = cl_bf16(x.detach()), cl_bf16(res)
    y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda", ldc=ops.row_pad(cout, torch.bfloat16), dtype=torch.bfloat16)
    rows = op.stats_rows(x_cl, y_cl)
    stats = torch.empty((rows, 2, cout), device="cuda")
    op.forward(x_cl, None, mod.bias.detach().cuda(), y_cl, stats=stats, add=r_cl)
    dw = torch.empty_like(wt)
    dbias = torch.empty(cout, device="cuda")
    op.wgrad(x_cl, None, cl(gy), dw, dbias)
    torch.cuda.synchronize()
    assert y_cl.dtype == torch.bfloat16
    got = y_cl.float().permute(0, 4, 1, 2, 3).cpu()
    err = (got - y_ref.detach()).abs().max().item() / y_ref.abs().max().item()
    assert err <= 1.5e-2, f"bf16-stored forward: {err:.3e}"
    werr = (dw.cpu() - mod.weight.grad).abs().max().item() / mod.weight.grad.abs().max().item()
    assert werr <= 1.5e-2, f"weight gradient from bf16-stored x: {werr:.3e}"
    close("bias gradient", dbias, mod.bias.grad)
    # statistics describe the fp32 values BEFORE the bf16 rounding of the store: compare with the stored tensor loosely
    st = stats.view(n, rows // n, 2, cout).double().sum(1).cpu()
    got_sum = got.double().sum(dim=(2, 3, 4))
    assert torch.allclose(st[:, 0], got_sum, rtol=2e-2, atol=2e-2 * max(1.0, got.abs().sum(dim=(2, 3, 4)).max().item()) ** 0.5 + 1.0)


@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", BF16_CASES)
def test_conv_bf16_operands(cin, cout, k, stride, transposed, shape):
    """dtype=BF16: operands rounded to bf16 (8 significant bits), fp32 accumulation.  Against the fp32 torch
    result the error of a K-term dot product is ~2^-8 * rms(terms) * sqrt(K) / ... : stated bound
    |err| <= 1.5e-2 * max|ref| (forward, input gradient and weight gradient)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(99 + cin + cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = torch.randn(n, cin, d, h, w, requires_grad=True)
    y_ref = mod(x)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
    wt = mod.weight.detach().cuda().contiguous()
    op.pack(wt)
    x_cl, gy_cl = cl(x.detach()), cl(gy)
    y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda")
    rows = op.stats_rows(x_cl, y_cl)
    stats = torch.empty((rows, 2, cout), device="cuda")
    op.forward(x_cl, None, mod.bias.detach().cuda(), y_cl, stats=stats)
    dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
    op.dgrad(gy_cl, dx_cl)
    dw = torch.empty_like(wt)
    dbias = torch.empty(cout, device="cuda")
    op.wgrad(x_cl, None, gy_cl, dw, dbias)
    torch.cuda.synchronize()

    def bf_close(name, got, ref):
        err = (got.cpu() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 1.5e-2 * scale + 1e-5, f"{name}: max|err|={err:.3e} (max|ref|={scale:.3e})"
        return err / scale

    e1 = bf_close("bf16 forward", ncdhw(y_cl), y_ref.detach())
    e2 = bf_close("bf16 dgrad", ncdhw(dx_cl), x.grad)
    if cin >= 16 and cout > 4:
        assert e1 > 1e-5, "bf16 path not taken (result is fp32-exact)"
    # 27-tap weight gradients also take bf16 operands (fp32 accumulation over voxels); 1x1x1 and the
    # small-channel paths stay fp32.  The bias gradient is summed from the fp32 values in every mode.
    bf_close("bf16 wgrad", dw, mod.weight.grad)
    close("bias gradient stays fp32", dbias, mod.bias.grad)
    st = stats.view(n, rows // n, 2, cout).double().sum(1).cpu()
    got_sum = ncdhw(y_cl).double().sum(dim=(2, 3, 4))
    assert torch.allclose(st[:, 0], got_sum, rtol=1e-3, atol=1e-2 * max(1.0, got_sum.abs().max().item()))


def test_batched_pack_matches_per_layer_pack():
    """One launch repacks every image of a model; each image must be bit-identical to the per-layer pack
    (fp32 and bf16 images, both orientations, 1x1, transposed, direct-path layers, ragged channel counts)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(5)
    layers = [(32, 64, 3, 2, False), (33, 40, 3, 1, False), (64, 3, 3, 2, True), (3, 3, 3, 1, False),
              (256, 512, 1, 1, False), (768, 128, 3, 2, True), (4, 32, 3, 2, False), (32, 3, 1, 1, False)]
    for dtype in (ops.F32, ops.BF16):
        convs, items = [], []
        for cin, cout, k, s, tr in layers:
            op = ops.ConvOp(cin, cout, k, s, tr, "cuda", dtype=dtype)
            shape = (cin, cout, k, k, k) if tr else (cout, cin, k, k, k)
            w = torch.randn(shape, device="cuda")
            op.pack(w)
            convs.append((op, w, op.packed_fwd.clone(), op.packed_dgrad.clone()))
            op.packed_fwd.fill_(0xAB)
            op.packed_dgrad.fill_(0xAB)
            items += [(op.d_fwd, w, op.packed_fwd), (op.d_dgrad, w, op.packed_dgrad)]
        ops.BatchedPacker(items, torch.device("cuda")).run()
        torch.cuda.synchronize()
        for (op, w, pf, pd), (cin, cout, k, s, tr) in zip(convs, layers):
            assert torch.equal(op.packed_fwd, pf), f"fwd image differs: {(cin, cout, k, s, tr)} dtype {dtype}"
            assert torch.equal(op.packed_dgrad, pd), f"dgrad image differs: {(cin, cout, k, s, tr)} dtype {dtype}"



TR_CASES = [c for c in BF16_CASES if c[2] == 3 and min(c[0], c[1]) > 4] + [
    (40, 72, 3, 1, False, (2, 5, 9, 11)),       # ragged tiles on every axis, channel counts that are no multiple of 32
    (32, 64, 3, 2, False, (1, 7, 9, 13)),       # stride 2 on odd extents
    (64, 32, 3, 2, True, (1, 3, 5, 7)),
]


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", TR_CASES)
def test_transposed_read_wgrad(cin, cout, k, stride, transposed, shape, stored):
    """MMTTA_OPT_WGRAD_VECTOR_STAGING = 1 (default): the weight gradient whose operands sit in LDS as [voxel][channel] and
    are transposed by ds_read_b64_tr_b16 on the way into the MFMA, against the fp32-operand kernel (option 0, also the
    fallback for operand pairs the transposed-read loader cannot take): within the bf16-operand bound of it and of torch
    fp32, bias gradient fp32-exact, with the norm-on-load of the module input and the accumulate path."""
    from multimodal_tta_amd import ops

    torch.manual_seed(31 + cin + 3 * cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.25).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None]).requires_grad_(True)
    y_ref = mod(xin)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    wt = mod.weight.detach().cuda().contiguous()
    x_cl = cl_bf16(x) if stored == "bf16" else cl(x)
    gy_cl = cl(gy)
    out = {}
    for mode in (0, 1):
        prev = ops.set_option(11, mode)
        try:
            op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
            op.pack(wt)
            dw = torch.empty_like(wt)
            db = torch.empty(cout, device="cuda")
            op.wgrad(x_cl, nl, gy_cl, dw, db)
            op.wgrad(x_cl, nl, gy_cl, dw, db, accumulate=True)
            torch.cuda.synchronize()
            out[mode] = (dw.cpu() / 2, db.cpu() / 2)
        finally:
            ops.set_option(11, prev)
    ref = mod.weight.grad
    scale = ref.abs().max().item()
    assert (out[1][0] - ref).abs().max().item() <= 1.5e-2 * scale + 1e-5
    assert (out[0][0] - ref).abs().max().item() <= 2e-4 * scale + 1e-5, "the fp32-operand kernel is the exact one"
    assert (out[1][0] - out[0][0]).abs().max().item() <= 1.5e-2 * scale + 1e-6, "differs from the fp32-operand kernel"
    close("bias gradient", out[1][1], mod.bias.grad)


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(32, 32, 3, 1, False, (1, 16, 16, 16)), (33, 32, 3, 1, False, (1, 8, 8, 16)),
                                  (40, 72, 3, 1, False, (2, 5, 9, 11)), (128, 136, 3, 1, False, (2, 4, 6, 8)),
                                  (512, 512, 3, 1, False, (1, 4, 4, 4)), (64, 64, 3, 1, False, (1, 8, 8, 8))])
def test_row_loader_matches_the_generic_loader(case, stored):
    """MMTTA_OPT_IGEMM_PIPELINE: the row-structured loader of the bf16 3x3x3 stride-1 stages (8-channel items, geometry
    once per tile, coefficients in LDS, ReLU as max(x, lo)) stages the same bf16 image as the generic walk: forward (with
    norm-on-load and ReLU, fp32- and bf16-stored input, channel counts that are no multiple of the stage) and input
    gradient are equal bit for bit, the statistics rows too."""
    from multimodal_tta_amd import ops

    cin, cout, k, stride, transposed, shape = case
    n, d, h, w = shape
    torch.manual_seed(11)
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 2 + 0.5).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    outs = {}
    for mode in (0, 1):
        prev = ops.set_option(6, mode)
        try:
            op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
            op.pack(mod.weight.detach().cuda().contiguous())
            x_cl = cl_bf16(x) if stored == "bf16" else cl(x)
            if stored == "bf16":       # input, output and fused add share one storage type
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda", ldc=ops.row_pad(cout, torch.bfloat16), dtype=torch.bfloat16)
            else:
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda")
            rows = op.stats_rows(x_cl, y_cl)
            stats = torch.zeros((rows, 2, cout), device="cuda")
            op.forward(x_cl, nl, mod.bias.detach().cuda(), y_cl, stats=stats)
            gy_cl = cl(torch.randn(n, cout, *y_cl.shape[1:4], generator=torch.Generator().manual_seed(5)))
            dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
            op.dgrad(gy_cl, dx_cl)
            torch.cuda.synchronize()
            outs[mode] = (y_cl.clone(), dx_cl.clone(), stats.clone())
        finally:
            ops.set_option(6, prev)
    assert torch.isfinite(outs[1][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]), "forward differs"
    assert torch.equal(outs[0][1], outs[1][1]), "input gradient differs"
    assert torch.equal(outs[0][2], outs[1][2]), "statistics differ"


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(32, 32, 3, 1, False, (1, 16, 16, 16)), (33, 32, 3, 1, False, (1, 8, 8, 16)),
                                  (16, 32, 3, 2, False, (1, 8, 8, 8)), (32, 64, 3, 2, False, (1, 8, 8, 16)),
                                  (128, 32, 3, 2, True, (1, 4, 4, 4)), (32, 32, 1, 1, False, (1, 9, 10, 11))])
def test_lean_tile_matches_the_wide_tile(case, stored):
    """MMTTA_OPT_IGEMM_LEAN: the 32-output-channel stride-1 forms of bf16 precision on the 4x8x8 tile (config 14, the
    default) against the 8x8x8 tile (config 7, option 10 = 0): forward with norm-on-load + ReLU and input gradient (the
    stride-2 layers reach it through their per-class input gradient / transposed forward) - the same products in the
    same order (up to the split of the reduction on small grids); the statistics rows are per tile and are compared as sums."""
    from multimodal_tta_amd import ops

    cin, cout, k, stride, transposed, shape = case
    n, d, h, w = shape
    torch.manual_seed(13)
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 2 + 0.5).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    outs, cfgs = {}, {}
    for mode in (0, 1):
        prev = ops.set_option(10, mode)
        prev_cls = ops.set_option(12, 0)          # per-class launches of the stride-2 forms (the class-fused kernel has its own tile)
        try:
            op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
            op.pack(mod.weight.detach().cuda().contiguous())
            x_cl = cl_bf16(x) if stored == "bf16" else cl(x)
            if stored == "bf16":
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda", ldc=ops.row_pad(cout, torch.bfloat16), dtype=torch.bfloat16)
            else:
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda")
            rows = op.stats_rows(x_cl, y_cl)
            stats = torch.zeros((rows, 2, cout), device="cuda")
            op.forward(x_cl, nl, mod.bias.detach().cuda(), y_cl, stats=stats)
            gy_cl = cl(torch.randn(n, cout, *y_cl.shape[1:4], generator=torch.Generator().manual_seed(5)))
            dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
            op.dgrad(gy_cl, dx_cl)
            torch.cuda.synchronize()
            cfgs[mode] = (int(op.plan(op.d_fwd, x_cl, y_cl).config), int(op.plan(op.d_dgrad, gy_cl, dx_cl).config))
            outs[mode] = (y_cl.clone(), dx_cl.clone(), stats.double().sum(0).cpu())
        finally:
            ops.set_option(12, prev_cls)
            ops.set_option(10, prev)
    assert 14 in cfgs[1] and 14 not in cfgs[0] and 7 in cfgs[0], f"plans {cfgs}"
    assert torch.isfinite(outs[1][0].float()).all()
    # twice the tiles can mean another split of the reduction (split-K below 96 workgroups): fp32 summation order only
    tol = 1e-2 if stored == "bf16" else 1e-4
    for i, what in ((0, "forward"), (1, "input gradient")):
        a, b = outs[0][i].float(), outs[1][i].float()
        assert (a - b).abs().max().item() <= (tol if i == 0 else 1e-4) * a.abs().max().item(), f"{what} differs"
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-3, atol=1e-2), "statistics sums differ"


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(64, 32, 3, 2, True, (1, 4, 4, 8)), (768, 128, 3, 2, True, (1, 8, 8, 8)),
                                  (128, 32, 3, 2, True, (2, 5, 6, 7)), (32, 64, 3, 2, False, (1, 8, 8, 16)),
                                  (64, 128, 3, 2, False, (1, 5, 6, 7)), (40, 72, 3, 2, True, (1, 3, 5, 9))])
def test_class_fused_stride2_forms_match_the_per_class_kernel(case, stored):
    """MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS: ConvTranspose3d forward and the input gradient of a stride-2 Conv3d with all
    8 output parity classes produced by one workgroup per coarse tile (forced here with a threshold of 1) against the
    per-class launches (threshold 0): same bf16 products, another fp32 summation order -> 1e-3 of max|ref| (bf16-stored
    outputs: one more rounding, 1e-2), statistics sums equal to 1e-3, and both within the bf16 bound of torch fp32."""
    from multimodal_tta_amd import ops

    cin, cout, k, stride, transposed, shape = case
    n, d, h, w = shape
    torch.manual_seed(17)
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.3).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None]).requires_grad_(True)
    y_ref = mod(xin)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    outs = {}
    for thr in (0, 1):
        prev = ops.set_option(12, thr)
        try:
            op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
            op.pack(mod.weight.detach().cuda().contiguous())
            bf = stored == "bf16" and transposed            # the forward of a transposed module may be bf16-stored
            x_cl = cl_bf16(x) if bf else cl(x)
            if bf:
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda", ldc=ops.row_pad(cout, torch.bfloat16), dtype=torch.bfloat16)
            else:
                y_cl = ops.new_cl(*op.out_shape(x_cl)[:4], cout, "cuda")
            rows = op.stats_rows(x_cl, y_cl)
            stats = torch.zeros((rows, 2, cout), device="cuda")
            op.forward(x_cl, nl, mod.bias.detach().cuda(), y_cl, stats=stats)
            dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
            op.dgrad(cl(gy), dx_cl)
            torch.cuda.synchronize()
            outs[thr] = (y_cl.float().clone(), dx_cl.clone(), stats.view(n, rows // n, 2, cout).double().sum(1).cpu())
        finally:
            ops.set_option(12, prev)
    ys, gs = y_ref.abs().max().item(), xin.grad.abs().max().item()
    ytol = 1e-2 if (stored == "bf16" and transposed) else 1e-3
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= ytol * ys, "forward differs from the per-class kernel"
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-3 * gs, "input gradient differs from the per-class kernel"
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-3, atol=1e-3 * max(1.0, outs[0][2].abs().max().item()))
    got_y = outs[1][0].permute(0, 4, 1, 2, 3).cpu()
    assert (got_y - y_ref.detach()).abs().max().item() <= 1.5e-2 * ys + 1e-5
    # dgrad here is the gradient w.r.t. the module input AFTER norm-on-load (the transform's backward is another kernel)
    got_dx = outs[1][1].permute(0, 4, 1, 2, 3).cpu()
    assert (got_dx - xin.grad).abs().max().item() <= 1.5e-2 * gs + 1e-5


@pytest.mark.parametrize("case", [(3, 3, (1, 5, 7, 70)), (4, 2, (2, 4, 9, 66)), (2, 4, (1, 3, 8, 128)), (1, 3, (1, 2, 3, 5))])
def test_thin_conv_on_matrix_tiles_matches_the_vector_kernel(case):
    """MMTTA_OPT_THIN_MFMA: the <= 4 -> <= 4 channel 3x3x3 convolution on the 4x4x4 matrix tiles (bf16 operands) against
    the fp32 vector-ALU kernel, with everything the epilogue can do: norm-on-load + ReLU of the input, bias, fused
    residual add with its own norm-on-load, accumulate, statistics.  Bound 1e-2 of max|ref| (operand rounding 2^-9)."""
    from multimodal_tta_amd import ops

    cin, cout, shape = case
    n, d, h, w = shape
    torch.manual_seed(23)
    mod = ref_module(cin, cout, 3, 1, False)
    x = torch.randn(n, cin, d, h, w) * 2 + 0.5
    r = torch.randn(n, cout, d, h, w)

    def stats_of(t):
        mu = t.mean(dim=(2, 3, 4))
        var = t.var(dim=(2, 3, 4), unbiased=False)
        return mu.reshape(-1).cuda().contiguous(), (1.0 / torch.sqrt(var + 1e-5)).reshape(-1).cuda().contiguous()

    mx, rx = stats_of(x)
    mr, rr = stats_of(r)
    gy = torch.randn(n, cout, d, h, w)
    outs = {}
    for mode in (0, 1, 2, 3):                     # vector kernel; 8 / 4 / 2 x 8 x 64 workgroup tiles of the matrix-tile kernel
        prev = ops.set_option(13, mode)
        try:
            op = ops.ConvOp(cin, cout, 3, 1, False, "cuda", dtype=ops.BF16)
            op.pack(mod.weight.detach().cuda().contiguous())
            x_cl, r_cl = cl(x), cl(r)
            y_cl = ops.new_cl(n, d, h, w, cout, "cuda")
            rows = op.stats_rows(x_cl, y_cl)
            stats = torch.zeros((rows, 2, cout), device="cuda")
            op.forward(x_cl, ops.NL(mx, rx, relu=True), mod.bias.detach().cuda(), y_cl, stats=stats, add=r_cl,
                       add_nl=ops.NL(mr, rr, relu=True))
            dx_cl = ops.new_cl(n, d, h, w, cin, "cuda")
            dx_cl.fill_(0.25)
            op.dgrad(cl(gy), dx_cl, accumulate=True)
            torch.cuda.synchronize()
            outs[mode] = (ncdhw(y_cl).clone(), ncdhw(dx_cl).clone(), stats.view(n, rows // n, 2, cout).double().sum(1).cpu())
        finally:
            ops.set_option(13, prev)
    ys, gs = outs[0][0].abs().max().item(), outs[0][1].abs().max().item()
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-2 * ys, "forward differs from the vector kernel"
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-2 * gs, "input gradient differs from the vector kernel"
    assert torch.allclose(outs[0][2], outs[1][2], rtol=2e-2, atol=2e-2 * max(1.0, outs[0][2].abs().max().item()))
    for mode in (2, 3):                           # the tile height changes which workgroup computes a voxel, not its value
        assert torch.equal(outs[mode][0], outs[1][0]) and torch.equal(outs[mode][1], outs[1][1]), f"tile mode {mode}"
        assert torch.allclose(outs[mode][2], outs[1][2], rtol=1e-5, atol=1e-5 * max(1.0, outs[1][2].abs().max().item()))
    y_ref = mod(F.relu(F.instance_norm(x))) + F.relu(F.instance_norm(r))
    assert (outs[1][0] - y_ref.detach()).abs().max().item() <= 1.5e-2 * y_ref.abs().max().item()


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(33, 32, (1, 5, 9, 11)), (96, 64, (2, 8, 8, 8)), (256, 512, (1, 4, 4, 4)), (40, 72, (1, 4, 16, 24)),
                                  (32, 3, (1, 6, 10, 70)), (64, 2, (2, 4, 8, 8))])      # heads: N <= 4 padded to one 32-column block
def test_pointwise_conv_weight_gradient_on_transposed_reads(case, stored):
    """1x1x1 weight gradient of bf16 precision (wgrad_tr1_kernel: a streaming kernel, both operands [voxel][channel] bf16 in
    LDS, one slab per wave) against torch fp32 - operands rounded to bf16: 1.5e-2 of max|ref| - and against the fp32-MFMA
    kernel it replaces (option 11 = 0): 1e-2; bias gradient fp32-exact; norm-on-load of the input, accumulate."""
    from multimodal_tta_amd import ops

    cin, cout, shape = case
    n, d, h, w = shape
    torch.manual_seed(41 + cin)
    mod = ref_module(cin, cout, 1, 1, False)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.25).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None])
    y_ref = mod(xin)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    wt = mod.weight.detach().cuda().contiguous()
    x_cl = cl_bf16(x) if stored == "bf16" else cl(x)
    gy_cl = cl(gy)
    out = {}
    for mode in (0, 1):
        prev = ops.set_option(11, mode)
        try:
            op = ops.ConvOp(cin, cout, 1, 1, False, "cuda", dtype=ops.BF16)
            op.pack(wt)
            dw = torch.empty_like(wt)
            db = torch.empty(cout, device="cuda")
            op.wgrad(x_cl, nl, gy_cl, dw, db)
            op.wgrad(x_cl, nl, gy_cl, dw, db, accumulate=True)
            torch.cuda.synchronize()
            out[mode] = (dw.cpu() / 2, db.cpu() / 2)
        finally:
            ops.set_option(11, prev)
    ref = mod.weight.grad
    scale = ref.abs().max().item()
    assert (out[1][0] - ref).abs().max().item() <= 1.5e-2 * scale + 1e-5
    assert (out[1][0] - out[0][0]).abs().max().item() <= 1e-2 * scale + 1e-6, "differs from the fp32-MFMA kernel"
    close("bias gradient", out[1][1], mod.bias.grad)


@pytest.mark.parametrize("r,cout,shape", [(3, 32, (1, 5, 6, 7)), (1, 64, (2, 4, 4, 8)), (4, 32, (1, 3, 3, 130))])
def test_head_input_gradient_into_a_bf16_stored_gradient(r, cout, shape):
    """The input gradient of a 1x1x1 head (R <= 4 -> C channels: pointwise_small_k_kernel, fp32 arithmetic) written into a
    bf16-stored gradient tensor (method.grad_storage in the deep-fusion decoder): round_bf16 of the fp32-stored result,
    accumulate in the gradient's storage."""
    from multimodal_tta_amd import ops

    torch.manual_seed(3 + r + cout)
    n, d, h, w = shape
    mod = ref_module(cout, r, 1, 1, False)
    wt = mod.weight.detach().cuda().contiguous()
    op = ops.ConvOp(cout, r, 1, 1, False, "cuda", dtype=ops.BF16)
    op.pack(wt)
    gy = torch.randn(n, r, d, h, w)
    want = torch.nn.functional.conv_transpose3d(gy, mod.weight.detach())          # dx of a 1x1x1 convolution
    res = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        dx = ops.new_cl(n, d, h, w, cout, "cuda", ldc=ops.row_pad(cout, dt), dtype=dt, zero=True)
        op.dgrad(cl(gy), dx)
        op.dgrad(cl(gy), dx, accumulate=True)
        torch.cuda.synchronize()
        res[name] = dx.float().permute(0, 4, 1, 2, 3).cpu() / 2
    scale = want.abs().max().item()
    assert (res["fp32"] - want).abs().max().item() <= 2e-5 * scale + 1e-6
    assert (res["bf16"] - want).abs().max().item() <= 3 * 2.0 ** -8 * scale


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,cout,shape", [(33, 32, (1, 16, 16, 17)), (32, 33, (2, 16, 16, 16)), (96, 64, (1, 8, 16, 33)),
                                            (64, 64, (1, 16, 16, 16)), (128, 40, (1, 16, 16, 16)), (16, 32, (1, 16, 16, 16))])
def test_streaming_pointwise_convolution(cin, cout, shape, stored):
    """1x1x1 convolutions over >= 4096 voxels in bf16 precision take the streaming matrix-core kernel (pointwise_mfma_kernel:
    A fragments loaded straight from HBM, no LDS): forward with bias and a fused add carrying its own norm-on-load (the
    shortcut convolution of a residual unit), and the input gradient with the accumulate path - against torch fp32 within the
    bf16-operand bound, ragged voxel counts, channel counts that are no multiple of 8 / 32, two batch items, both storages."""
    from multimodal_tta_amd import ops

    torch.manual_seed(23 + cin + cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, 1, 1, False)
    x = torch.randn(n, cin, d, h, w).to(torch.bfloat16).float().requires_grad_(True)
    other = (torch.randn(n, cout, d, h, w) * 1.5 + 0.2).to(torch.bfloat16).float()
    mu = other.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(other.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    y_ref = mod(x) + F.relu((other - mu[:, :, None, None, None]) * rstd[:, :, None, None, None])
    gy = torch.randn_like(y_ref).to(torch.bfloat16).float()
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    conv = cl_bf16 if stored == "bf16" else cl
    dt = torch.bfloat16 if stored == "bf16" else torch.float32
    op = ops.ConvOp(cin, cout, 1, 1, False, "cuda", dtype=ops.BF16)
    op.pack(mod.weight.detach().cuda().contiguous())
    y = ops.new_cl(n, d, h, w, cout, "cuda", ldc=ops.row_pad(cout, dt), dtype=dt, zero=True)
    op.forward(conv(x.detach()), None, mod.bias.detach().cuda(), y, add=conv(other), add_nl=nl)
    dx = ops.new_cl(n, d, h, w, cin, "cuda", ldc=ops.row_pad(cin, dt), dtype=dt, zero=True)
    op.dgrad(conv(gy), dx)
    op.dgrad(conv(gy), dx, accumulate=True)
    torch.cuda.synchronize()
    got_y = y.float().permute(0, 4, 1, 2, 3).cpu()
    got_dx = dx.float().permute(0, 4, 1, 2, 3).cpu() / 2
    assert (got_y - y_ref.detach()).abs().max().item() <= 1.5e-2 * y_ref.abs().max().item()
    assert (got_dx - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()


def cl_thin_bf16(x):
    """NCDHW cpu fp32 (<= 4 channels) -> bf16-stored channels-last cuda view with 8-byte voxels."""
    from multimodal_tta_amd import ops
    n, c, d, h, w = x.shape
    out = ops.new_cl(n, d, h, w, c, "cuda", ldc=4, zero=True, dtype=torch.bfloat16)
    ops.to_cl(x.cuda().contiguous(), out=out)          # mmtta_copy_strided rounds on the way (round to nearest even)
    return out


@pytest.mark.parametrize("with_norm", [False, True])
@pytest.mark.parametrize("cin,stride,shape", [(4, 2, (1, 16, 16, 16)), (2, 2, (2, 6, 10, 12)), (3, 1, (1, 5, 7, 11)), (4, 1, (1, 4, 9, 8))])
def test_bf16_stored_network_input_gives_the_same_bits(cin, stride, shape, with_norm):
    """The staged network input of bf16 precision is bf16-stored with 8-byte voxels (models/unet.py::input_dtype): the thin-K
    forward (chan_mfma_kernel) and the thin weight gradient (wgrad_thin_tr_kernel) round their gathered operand to bf16 while
    staging, so a bf16-representable input gives the SAME BITS from either storage - output, statistics, weight and bias
    gradient; with a norm-on-load in front the fp32 transform is applied to the same values.  mmtta_copy_strided into a
    bf16 destination rounds to nearest even."""
    from multimodal_tta_amd import ops

    torch.manual_seed(40 + cin + stride)
    n, d, h, w = shape
    cout = 32
    mod = ref_module(cin, cout, 3, stride, False)
    x_raw = torch.randn(n, cin, d, h, w) * 1.5 + 0.25
    x = x_raw.to(torch.bfloat16).float()
    x16 = cl_thin_bf16(x_raw)
    assert torch.equal(x16.float().permute(0, 4, 1, 2, 3).cpu(), x), "copy_strided -> bf16 is round-to-nearest-even"
    nl = None
    if with_norm:
        mu = x.mean(dim=(2, 3, 4))
        rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
        nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    op = ops.ConvOp(cin, cout, 3, stride, False, "cuda", dtype=ops.BF16)
    wt = mod.weight.detach().cuda().contiguous()
    op.pack(wt)
    gy = None
    res = {}
    for name, xin in (("fp32", cl(x)), ("bf16", x16)):
        y = ops.new_cl(*op.out_shape(xin)[:4], cout, "cuda", ldc=ops.row_pad(cout, torch.bfloat16), dtype=torch.bfloat16)
        rows = op.stats_rows(xin, y)
        stats = torch.empty((rows, 2, cout), device="cuda")
        op.forward(xin, nl, mod.bias.detach().cuda(), y, stats=stats)
        if gy is None:
            gy = torch.randn(n, cout, *y.shape[1:4]).to(torch.bfloat16).float()
        dw = torch.empty_like(wt)
        db = torch.empty(cout, device="cuda")
        op.wgrad(xin, nl, cl_bf16(gy), dw, db)
        torch.cuda.synchronize()
        res[name] = (y.clone(), stats.clone(), dw.clone(), db.clone())
    for i, what in enumerate(("output", "statistics rows", "weight gradient", "bias gradient")):
        assert torch.equal(res["fp32"][i], res["bf16"][i]), f"{what} differs between the fp32- and the bf16-stored input"
    # and against torch (operand rounding bound of the bf16 kernels)
    xin_ref = x if nl is None else F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None])
    xin_ref = xin_ref.clone().requires_grad_(True)
    y_ref = mod(xin_ref)
    y_ref.backward(gy)
    got = res["bf16"][0].float().permute(0, 4, 1, 2, 3).cpu()
    assert (got - y_ref.detach()).abs().max().item() <= 1.5e-2 * y_ref.abs().max().item()
    assert (res["bf16"][2].cpu() - mod.weight.grad).abs().max().item() <= 1.5e-2 * mod.weight.grad.abs().max().item()


@pytest.mark.parametrize("r,shape", [(3, (1, 5, 7, 70)), (1, (2, 4, 6, 66)), (4, (1, 3, 9, 130)), (2, (1, 8, 8, 8))])
def test_thin_gradients_bf16_stored_give_the_same_bits(r, shape):
    """bf16-stored thin gradients (8-byte voxels; models/unet.py::thin_grad_dtype) around the head of the U-Net:
      * the input gradient of the R -> R convolution (conv3_mfma4_kernel) with d(logits), its identity-residual term and
        its result all bf16-stored = round_bf16 of the fp32-stored call;
      * its weight / bias gradient (wgrad_thin_tr_kernel, thin dense operand) from a bf16-stored d(logits): the same bits;
      * the up-convolution K -> R: input gradient (chan_mfma_kernel, 64-column form) and weight gradient
        (wgrad_thin_tr_kernel) from a bf16-stored output gradient: the same bits."""
    from multimodal_tta_amd import ops

    torch.manual_seed(60 + r)
    n, d, h, w = shape
    gy = torch.randn(n, r, d, h, w).to(torch.bfloat16).float()
    g32, g16 = cl(gy), cl_thin_bf16(gy)
    # R -> R convolution (k3 s1): dx = dgrad(dy) + dy (identity residual term fused into the epilogue)
    mod = ref_module(r, r, 3, 1, False)
    op = ops.ConvOp(r, r, 3, 1, False, "cuda", dtype=ops.BF16)
    wt = mod.weight.detach().cuda().contiguous()
    op.pack(wt)
    x = torch.randn(n, r, d, h, w)
    x_cl = cl(x)
    res = {}
    for name, g in (("fp32", g32), ("bf16", g16)):
        dx = ops.new_cl(n, d, h, w, r, "cuda", ldc=4, zero=True, dtype=g.dtype)
        op.dgrad(g, dx, add=g)
        dw = torch.empty_like(wt)
        db = torch.empty(r, device="cuda")
        op.wgrad(x_cl, None, g, dw, db)
        torch.cuda.synchronize()
        res[name] = (dx.float().clone(), dw.clone(), db.clone())
    assert torch.equal(res["bf16"][0], res["fp32"][0].to(torch.bfloat16).float()), "R -> R input gradient: not round_bf16 of the fp32-stored call"
    assert torch.equal(res["bf16"][1], res["fp32"][1]), "R -> R weight gradient differs"
    assert torch.equal(res["bf16"][2], res["fp32"][2]), "R -> R bias gradient differs"
    # up-convolution 64 -> R (k3 s2): gradients from the thin output gradient
    if d % 2 == 0 and h % 2 == 0 and w % 2 == 0:
        K = 64
        up = ref_module(K, r, 3, 2, True)
        opu = ops.ConvOp(K, r, 3, 2, True, "cuda", dtype=ops.BF16)
        wu = up.weight.detach().cuda().contiguous()
        opu.pack(wu)
        xc = (torch.randn(n, K, d // 2, h // 2, w // 2)).to(torch.bfloat16).float()
        xc16 = cl_bf16(xc)
        resu = {}
        for name, g in (("fp32", g32), ("bf16", g16)):
            dxc = ops.new_cl(n, d // 2, h // 2, w // 2, K, "cuda", ldc=ops.row_pad(K, torch.bfloat16), dtype=torch.bfloat16, zero=True)
            opu.dgrad(g, dxc)
            dw = torch.empty_like(wu)
            db = torch.empty(r, device="cuda")
            opu.wgrad(xc16, None, g, dw, db)
            torch.cuda.synchronize()
            resu[name] = (dxc.float().clone(), dw.clone(), db.clone())
        for i, what in enumerate(("input gradient", "weight gradient", "bias gradient")):
            assert torch.equal(resu["fp32"][i], resu["bf16"][i]), f"up-convolution {what} differs between the storages"


THIN_TR_CASES = [
    (4, 32, 3, 2, False, (1, 16, 16, 16)),      # first encoder layer: Q = x (4 channels, norm-on-load), P = dy, bias from P
    (3, 32, 3, 1, False, (1, 5, 7, 11)),        # stride 1, ragged tiles
    (1, 32, 3, 2, False, (2, 6, 10, 12)),       # one-channel stem, two batch items, odd coarse extents
    (2, 64, 3, 1, False, (1, 4, 9, 8)),         # two column blocks per workgroup with the bias partials
    (64, 3, 3, 2, True, (1, 3, 5, 70)),         # full-resolution up-convolution: Q = dy, P = x (norm-on-load), two column blocks
    (32, 2, 3, 2, True, (2, 4, 4, 16)),
    (64, 1, 3, 2, True, (1, 3, 4, 9)),
    (96, 4, 3, 2, True, (1, 2, 3, 5)),          # three column blocks: one per workgroup
    # <= 4 channels on BOTH sides (stride 1): the dense operand is a thin fp32 tensor as well
    (3, 3, 3, 1, False, (1, 5, 7, 70)),
    (4, 2, 3, 1, False, (2, 5, 6, 70)),
    (1, 1, 3, 1, False, (1, 6, 5, 9)),
    (2, 4, 3, 1, False, (1, 3, 9, 130)),
    (3, 3, 3, 1, False, (1, 8, 8, 8)),
]


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", THIN_TR_CASES)
def test_thin_wgrad_on_transposed_reads(cin, cout, k, stride, transposed, shape, stored):
    """bf16 precision, 27 taps, <= 4 channels on one side: wgrad_thin_tr_kernel (both operands bf16 rows in LDS, fragments by
    ds_read_b64_tr_b16, the taps folded into the transposed read's chunk addresses) against wgrad_small_kernel's bf16
    branch (MMTTA_OPT_WGRAD_VECTOR_STAGING = 0: the same bf16-rounded operands, fp32 accumulation in another order) and
    torch fp32 - with the norm-on-load of the module input, the accumulate path, the wide operand fp32- or bf16-stored.
    With <= 4 channels on both sides option 0 is the fp32 vector-ALU kernel (wgrad_tiny_kernel): the bf16-operand bound."""
    from multimodal_tta_amd import _lib, ops
    import ctypes as C

    tiny = cin <= 4 and cout <= 4
    if tiny and stored == "bf16":
        pytest.skip("thin tensors are fp32-stored")

    torch.manual_seed(5 + cin + 7 * cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.25).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None]).requires_grad_(True)
    y_ref = mod(xin)
    gy = torch.randn_like(y_ref).to(torch.bfloat16).float()
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    wt = mod.weight.detach().cuda().contiguous()
    wide_is_x = transposed
    x_cl = cl_bf16(x) if (stored == "bf16" and wide_is_x) else cl(x)
    gy_cl = cl_bf16(gy) if (stored == "bf16" and not wide_is_x) else cl(gy)
    out = {}
    for mode in (0, 1):
        prev = ops.set_option(11, mode)
        try:
            op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
            kid = int(_lib.load().mmtta_conv_wgrad_kernel(C.byref(op.d_fwd), C.byref(ops.desc_cl(x_cl)), C.byref(ops.desc_cl(gy_cl))))
            assert kid == (10 if mode else (6 if tiny else 3)), f"option 11 = {mode}: weight-gradient kernel {kid}"
            op.pack(wt)
            dw = torch.empty_like(wt)
            db = torch.empty(cout, device="cuda")
            op.wgrad(x_cl, nl, gy_cl, dw, db)
            op.wgrad(x_cl, nl, gy_cl, dw, db, accumulate=True)
            torch.cuda.synchronize()
            out[mode] = (dw.cpu() / 2, db.cpu() / 2)
        finally:
            ops.set_option(11, prev)
    ref = mod.weight.grad
    scale = ref.abs().max().item()
    assert (out[1][0] - ref).abs().max().item() <= 1.5e-2 * scale + 1e-5
    assert (out[1][0] - out[0][0]).abs().max().item() <= (1.5e-2 if tiny else 2e-4) * scale + 1e-6, \
        "differs from the other kernel beyond " + ("the bf16 rounding of the operands" if tiny else "summation order")
    close("bias gradient", out[1][1], mod.bias.grad)
    close("bias gradient (both kernels)", out[1][1], out[0][1])


@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", [
    (32, 64, 3, 2, False, (2, 16, 16, 32)), (64, 128, 3, 2, False, (1, 9, 11, 13)), (128, 32, 3, 2, True, (2, 6, 8, 8)),
    (256, 64, 3, 2, True, (1, 4, 4, 8)), (64, 64, 3, 1, False, (2, 8, 16, 16)), (128, 128, 3, 1, False, (1, 5, 9, 11)),
    (40, 72, 3, 2, False, (1, 8, 8, 16)), (32, 40, 3, 2, False, (2, 8, 8, 16)), (40, 32, 3, 2, True, (1, 4, 6, 8)),
    (24, 48, 3, 1, False, (1, 8, 8, 8))])
def test_paired_column_blocks_of_the_transposed_read_wgrad(cin, cout, k, stride, transposed, shape):
    """Both operands bf16-stored (method.storage + method.grad_storage: bf16): the stride-2 layers, and the stride-1 layers of
    <= 128 dense channels, stage the gathered box once for TWO 32-channel blocks of the dense operand (wgrad_tr_kernel NB =
    2, one workgroup per CU, twice the slabs).  Several tiles per slab, two batch items, ragged extents, the norm-on-load of
    the module input on either side (gathered: convolution, dense: transposed convolution), bias gradient, accumulate -
    against torch fp32 within the bf16-operand bound; (40, 72) has an odd number of column blocks and stays unpaired, 40 and
    48 dense channels make the second block of the pair a ragged one (8 / 16 live channels)."""
    from multimodal_tta_amd import ops

    torch.manual_seed(41 + cin + 3 * cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.25).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None]).requires_grad_(True)
    y_ref = mod(xin)
    gy = torch.randn_like(y_ref).to(torch.bfloat16).float()
    y_ref.backward(gy)
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    wt = mod.weight.detach().cuda().contiguous()
    op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
    op.pack(wt)
    import ctypes as C
    from multimodal_tta_amd import _lib
    kid = _lib.load().mmtta_conv_wgrad_kernel(C.byref(op.d_fwd), C.byref(ops.desc_cl(cl_bf16(x))), C.byref(ops.desc_cl(cl_bf16(gy))))
    assert kid in (7, 8), "the transposed-read kernel takes these layers"
    dw = torch.empty_like(wt)
    db = torch.empty(cout, device="cuda")
    op.wgrad(cl_bf16(x), nl, cl_bf16(gy), dw, db)
    op.wgrad(cl_bf16(x), nl, cl_bf16(gy), dw, db, accumulate=True)
    torch.cuda.synchronize()
    ref = mod.weight.grad
    assert (dw.cpu() / 2 - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item() + 1e-5
    close("bias gradient", db.cpu() / 2, mod.bias.grad)


@pytest.mark.parametrize("cin,cout,k,stride,transposed,shape", [
    (32, 32, 3, 1, False, (1, 8, 8, 16)), (64, 128, 3, 2, False, (1, 6, 6, 8)), (64, 32, 3, 2, True, (1, 4, 4, 8)),
    (256, 512, 1, 1, False, (1, 4, 4, 4)), (40, 72, 3, 1, False, (2, 5, 9, 11))])
def test_bf16_stored_gradients_through_the_convolutions(cin, cout, k, stride, transposed, shape):
    """method.grad_storage: bf16 - the output gradient dy arrives bf16-stored and the input gradient dx leaves bf16-stored.
    With a bf16-representable dy the weight / bias gradient equal the fp32-stored call BIT FOR BIT (the kernels round dy to
    bf16 while staging it anyway: same operands, same order - at these sizes also where the all-bf16 call pairs dense column
    blocks, wgrad_tr_kernel NB = 2: the slab count is capped by the tile count either way) and dx equals
    round_bf16(fp32-stored dx) to one bf16 ulp."""
    from multimodal_tta_amd import ops

    torch.manual_seed(7 + cin)
    n, d, h, w = shape
    mod = ref_module(cin, cout, k, stride, transposed)
    x = (torch.randn(n, cin, d, h, w) * 1.3).to(torch.bfloat16).float()
    x16 = cl_bf16(x)
    op = ops.ConvOp(cin, cout, k, stride, transposed, "cuda", dtype=ops.BF16)
    wt = mod.weight.detach().cuda().contiguous()
    op.pack(wt)
    _, do, ho, wo, _ = op.out_shape(x16)
    gy = torch.randn(n, cout, do, ho, wo).to(torch.bfloat16).float()
    res = {}
    for name, gcl in (("fp32", cl(gy)), ("bf16", cl_bf16(gy))):
        dw = torch.empty_like(wt)
        db = torch.empty(cout, device="cuda")
        op.wgrad(x16, None, gcl, dw, db)
        dx = ops.new_cl(n, d, h, w, cin, "cuda", ldc=ops.row_pad(cin, gcl.dtype), dtype=gcl.dtype, zero=True)
        op.dgrad(gcl, dx)
        op.dgrad(gcl, dx, accumulate=True, add=dx.clone())          # accumulate + fused add in the gradient's storage
        torch.cuda.synchronize()
        res[name] = (dw.clone(), db.clone(), dx.float().clone())
    assert torch.equal(res["fp32"][0], res["bf16"][0]), "weight gradient differs between fp32- and bf16-stored dy"
    assert torch.equal(res["fp32"][1], res["bf16"][1]), "bias gradient differs"
    want = res["fp32"][2]
    err = (res["bf16"][2] - want).abs().max().item() / want.abs().max().item()
    assert err <= 3 * 2.0 ** -8, f"bf16-stored input gradient: {err:.3e} of max (three roundings to bf16)"


@pytest.mark.parametrize("stored", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,cout,shape", [(64, 3, (1, 5, 6, 19)), (32, 1, (2, 4, 4, 8)), (64, 4, (1, 3, 9, 33)), (32, 3, (1, 8, 8, 16))])
def test_upconvolution_as_a_gather_gemm(cin, cout, shape, stored):
    """ConvTranspose3d K -> R (k3 s2, R <= 4) in bf16 precision (upconv8_kernel: the 27 taps regrouped as 8 coarse offsets x
    8 output parities, W' in the packed image): norm-on-load of a bf16- or fp32-stored input, bias, per-tile statistics,
    a fused residual operand and the accumulate path, ragged extents (tile borders on every axis), against torch fp32."""
    from multimodal_tta_amd import ops

    torch.manual_seed(17 + cin + cout)
    n, d, h, w = shape
    mod = ref_module(cin, cout, 3, 2, True)
    x = (torch.randn(n, cin, d, h, w) * 1.5 + 0.2).to(torch.bfloat16).float()
    mu = x.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    xin = F.relu((x - mu[:, :, None, None, None]) * rstd[:, :, None, None, None])
    res = torch.randn(n, cout, 2 * d, 2 * h, 2 * w)
    y_ref = mod(xin) + res
    op = ops.ConvOp(cin, cout, 3, 2, True, "cuda", dtype=ops.BF16)
    op.pack(mod.weight.detach().cuda().contiguous())
    nl = ops.NL(mu.reshape(-1).cuda().contiguous(), rstd.reshape(-1).cuda().contiguous(), relu=True)
    x_cl = cl_bf16(x) if stored == "bf16" else cl(x)
    y_cl = ops.new_cl(n, 2 * d, 2 * h, 2 * w, cout, "cuda", ldc=4, zero=True)
    rows = op.stats_rows(x_cl, y_cl)
    assert rows == n * ((d + 3) // 4) * ((h + 3) // 4) * ((w + 7) // 8), "one statistics row per coarse 4x4x8 tile"
    stats = torch.full((rows, 2, cout), float("nan"), device="cuda")
    bias = mod.bias.detach().cuda()
    op.forward(x_cl, nl, bias, y_cl, stats=stats, add=cl(res))
    torch.cuda.synchronize()
    got = ncdhw(y_cl)
    scale = y_ref.abs().max().item()
    err = (got - y_ref.detach()).abs().max().item() / scale
    assert err <= 1.5e-2, f"forward: {err:.3e}"
    st = stats.view(n, rows // n, 2, cout).double().sum(1).cpu()
    assert torch.allclose(st[:, 0], got.double().sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(st[:, 1], (got.double() ** 2).sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-2)
    # accumulate: y += conv(x) (no add): twice the convolution part
    y2 = ops.new_cl(n, 2 * d, 2 * h, 2 * w, cout, "cuda", ldc=4, zero=True)
    op.forward(x_cl, nl, bias, y2)
    op.forward(x_cl, nl, bias, y2, accumulate=True)
    torch.cuda.synchronize()
    want2 = 2.0 * mod(xin).detach()
    assert (ncdhw(y2) - want2).abs().max().item() / want2.abs().max().item() <= 1.5e-2
