"""Whole-network parity of the HIP deep-fusion U-Net (reference src/models/unet_multimodal_midfusion.py)
against the torch-CPU oracle: logits, every parameter gradient, missing-modality forward, adaptation loop."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(name="unet_multimodal_deepfusion", num_modalities=4, num_classes=3, spatial_dims=3,
           channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2], num_res_units=2, norm="INSTANCE", act="RELU", dropout=0.0)


def build_pair(cfg, seed=42):
    import oracle
    from multimodal_tta_amd.models import MultimodalUNetDeepFusion
    torch.manual_seed(seed)
    ref = oracle.MultimodalUNetDeepFusion(cfg)
    hip = MultimodalUNetDeepFusion(cfg)
    assert list(ref.state_dict().keys()) == list(hip.state_dict().keys())
    hip.load_state_dict(ref.state_dict())
    return ref, hip.cuda()


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def feeds_norm(model, pname):
    if not pname.endswith(".conv.bias"):
        return False
    parent = model.get_submodule(pname[:-len(".conv.bias")])
    adn = getattr(parent, "adn", None)
    return adn is not None and hasattr(adn, "N") and not isinstance(adn.N, torch.nn.GroupNorm)


def relu_margin(model64, x64):
    """Smallest |normalised pre-activation| anywhere in the network, evaluated in float64."""
    lo = [float("inf")]
    hooks = [m.register_forward_hook(lambda mod, i, o: lo.__setitem__(0, min(lo[0], o.detach().abs().min().item())))
             for m in model64.modules()
             if isinstance(m, (torch.nn.InstanceNorm3d, torch.nn.BatchNorm3d, torch.nn.GroupNorm))]
    with torch.no_grad():
        model64(x64)
    for h in hooks:
        h.remove()
    return lo[0]


def input_without_relu_ties(model64, shape, margin=5e-6, tries=48):
    """A seeded input for which no normalised activation sits within `margin` of the ReLU threshold.

    Two fp32 evaluations that sum in different orders (torch CPU vs the MFMA k-order) differ by ~1e-6 in a
    normalised activation; when that activation is 0 to rounding the two put it on different sides of the
    ReLU, and in networks this small one such flip moves every upstream gradient by ~1 % (measured: a
    pre-activation of -1.3e-6 in the oracle, +5e-8 on the GPU).  That is a property of comparing any two fp32
    implementations, not of either one, so the parity input is chosen - reproducibly - to have no such tie."""
    for seed in range(1, tries + 1):
        torch.manual_seed(seed)
        x = torch.randn(shape)
        if relu_margin(model64, x.double()) > margin:
            return x
    pytest.skip("no tie-free input found")


@pytest.mark.parametrize("cfg_over,shape", [({}, (1, 4, 32, 32, 32)),
                                            ({"num_modalities": 2, "num_classes": 1}, (1, 2, 16, 48, 48))])
def test_deepfusion_forward_backward_parity(cfg_over, shape):
    """Logits and every parameter gradient against the oracle (fp32, with the float64 oracle choosing a
    tie-free input, see input_without_relu_ties)."""
    cfg = dict(CFG, **cfg_over)
    ref, hip = build_pair(cfg)
    ref.train(); hip.train()
    x = input_without_relu_ties(copy.deepcopy(ref).double().train(), shape)
    z_ref, z_hip = ref(x), hip(x.cuda())
    e = (z_hip.cpu() - z_ref).abs().max().item() / z_ref.abs().max().item()
    assert e < 5e-4, f"logits rel err {e:.3e}"
    g = torch.randn_like(z_ref)
    (z_ref * g).sum().backward()
    (z_hip * g.cuda()).sum().backward()
    ref_grads = {n: p.grad for n, p in ref.named_parameters()}
    hip_params = dict(hip.named_parameters())
    worst = ("", 0.0)
    bad = []
    for name, gr in ref_grads.items():
        if gr is None:                              # domain_classifier: not on the segmentation path
            assert name.startswith("domain_classifier")
            continue
        gh = hip_params[name].grad
        assert gh is not None, name
        scale = gr.abs().max().item()
        if feeds_norm(ref, name):
            wscale = ref_grads[name[:-len("bias")] + "weight"].abs().max().item()
            assert gh.abs().max().item() <= 2e-3 * wscale + 1e-4, name
            continue
        err = (gh.cpu() - gr).abs().max().item()
        worst = max(worst, (name, err / (scale + 1e-6)), key=lambda t: t[1])
        if err > 2e-3 * scale + 2e-6:
            bad.append(f"{name}: grad err {err:.3e} vs scale {scale:.3e}")
    print("worst grad", worst)
    assert not bad, "\n".join(bad)


def test_deepfusion_missing_modality_forward_and_tta():
    import oracle
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.registry import get_plugin
    ref, hip = build_pair(CFG)
    x = torch.randn(1, 4, 32, 32, 32)
    present = [True, False, True, True]
    ref.eval(); hip.eval()
    with torch.no_grad():
        xm = x.clone(); xm[:, 1] = 0
        z_ref = ref(xm, present=present)
        z_hip = hip(xm.cuda(), present=present).cpu()
    assert (z_hip - z_ref).abs().max().item() / z_ref.abs().max().item() < 5e-4
    # adaptation with a missing modality + per-step modality dropout (BASELINE configs[4])
    cfg = compose(overrides=["task=brats", "model=unet_multimodal_deepfusion", "method=tta_moddrop"])
    cfg["model"] = dict(CFG)
    cfg["method"]["steps"] = 3
    cfg["method"]["moddrop"]["p"] = 0.5
    cfg["training"]["optimizers"]["adam"]["lr"] = 1e-4
    ref0 = copy.deepcopy(ref)
    out_ref = oracle.adapt_volume(ref, x, cfg["training"], steps=3, missing=[1], moddrop_p=0.5, moddrop_seed=0,
                                  masked_means=True)
    plug = get_plugin("entmin_tta")(cfg).setup(hip, "cuda")
    res = plug.adapt_volume(x.cuda())
    for a, b in zip(res["losses"].cpu().tolist(), out_ref["losses"]):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (a, b)
    m64 = copy.deepcopy(ref0).double()
    z64 = oracle.adapt_volume(m64, x.double(), cfg["training"], steps=3, missing=[1], moddrop_p=0.5, moddrop_seed=0,
                              masked_means=True)["logits"]
    scale = z64.abs().max().item()
    e_ref = (out_ref["logits"].double() - z64).abs().max().item() / scale
    e_hip = (plug.logits(res).cpu().double() - z64).abs().max().item() / scale
    # Adam moves every parameter whose gradient is rounding noise (biases in front of an InstanceNorm) by +-lr per
    # step in a direction that depends on the summation order, so after 3 steps at lr 1e-4 the logits of two fp32
    # implementations differ by a few 1e-3 of their range whatever the kernels do (measured 1.9e-3 .. 2.1e-3 across
    # builds of this repository); the losses above are the tight check.
    assert e_hip <= max(4e-3, 3 * e_ref), (e_hip, e_ref)


@pytest.mark.parametrize("over,shape", [({}, (2, 4, 32, 32, 32)), ({"num_modalities": 2, "num_classes": 1}, (1, 2, 32, 32, 64))])
def test_auxiliary_outputs_match_the_oracle(over, shape):
    """``forward(..., return_intermediate_features=True)`` and ``return_domain_logits=True`` (reference
    src/models/unet_multimodal_midfusion.py:204-209,258-265): same tuple structure, shared / specific global means
    and the domain classifier's logits against the oracle restatement; disabled classifier -> plain logits."""
    cfg = dict(CFG, **over)
    ref, hip = build_pair(cfg)
    torch.manual_seed(1)
    x = torch.randn(shape)
    ref.eval()
    hip.eval()
    with torch.no_grad():
        z_ref, sh_ref, sp_ref = ref(x, return_intermediate_features=True)
        z2_ref, dl_ref = ref(x, return_domain_logits=True)
        z, sh, sp = hip(x.cuda(), return_intermediate_features=True)
        z2, dl = hip(x.cuda(), return_domain_logits=True)
    M = cfg["num_modalities"]
    assert len(sh) == len(sh_ref) == M and len(sp) == len(sp_ref) == M
    assert rel_err(z, z_ref) < 5e-4 and rel_err(z2, z2_ref) < 5e-4
    for a, b in zip(sh + sp, sh_ref + sp_ref):
        assert a.shape == b.shape and rel_err(a, b) < 1e-4, (a.shape, b.shape, rel_err(a, b))
    assert dl.shape == dl_ref.shape == (shape[0] * M, M)
    assert rel_err(dl, dl_ref) < 2e-4, rel_err(dl, dl_ref)
    # the first flag wins when both are set; a model without the classifier returns plain logits (reference :258-265)
    with torch.no_grad():
        both = hip(x.cuda(), return_domain_logits=True, return_intermediate_features=True)
    assert len(both) == 3
    ref2, hip2 = build_pair(dict(cfg, domain_classifier={"enabled": False}))
    with torch.no_grad():
        plain = hip2(x.cuda(), return_domain_logits=True)
    assert torch.is_tensor(plain) and plain.shape == z_ref.shape
    assert hip2.get_domain_loss_weight() == 0.0 and hip.get_domain_loss_weight() == 0.1
