"""Whole-network parity of the HIP ``unet`` against the torch-CPU oracle on identical weights and
inputs: logits, input-independent gradients of every parameter, state_dict key compatibility.

Tolerance (fp32 path): logits within 5e-4 * max|logits|; gradients within 2e-3 relative to the
largest gradient entry of the same tensor (+1e-6): backward chains ~40 kernels whose fp32
reductions run in a different order than oneDNN's.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(in_channels=4, num_classes=3, spatial_dims=3, channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2],
             num_res_units=2, norm="INSTANCE", act="RELU", dropout=0.0)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def feeds_norm(model, pname: str) -> bool:
    """True for '<Convolution>.conv.bias' when that Convolution has an ADN with a norm layer."""
    if not pname.endswith(".conv.bias"):
        return False
    parent = model.get_submodule(pname[:-len(".conv.bias")])
    adn = getattr(parent, "adn", None)
    # instance / batch norm subtract the per-channel mean; a group norm's mean spans several channels
    return adn is not None and hasattr(adn, "N") and not isinstance(adn.N, torch.nn.GroupNorm)


def build_pair(cfg, seed=42):
    import oracle
    from multimodal_tta_amd.models import UNet

    torch.manual_seed(seed)
    ref = oracle.UNet(cfg)
    hip = UNet(cfg)
    assert list(ref.state_dict().keys()) == list(hip.state_dict().keys())
    for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), hip.state_dict().items()):
        assert v1.shape == v2.shape, k1
    hip.load_state_dict(ref.state_dict())
    return ref, hip.cuda()


@pytest.mark.parametrize("cfg_over,shape", [
    ({}, (1, 4, 32, 32, 32)),
    ({"in_channels": 2, "num_classes": 1}, (1, 2, 16, 48, 48)),
    ({"norm": "BATCH"}, (2, 4, 16, 16, 16)),
    ({"norm": ("GROUP", {"num_groups": 2}), "num_classes": 4}, (1, 4, 16, 16, 32)),
    ({"num_res_units": 0, "norm": "BATCH"}, (2, 4, 16, 16, 16)),
])
def test_unet_forward_backward_parity(cfg_over, shape):
    cfg = dict(SMALL, **cfg_over)
    ref, hip = build_pair(cfg)
    torch.manual_seed(0)
    x = torch.randn(shape)
    ref.train()
    hip.train()
    z_ref = ref(x)
    z_hip = hip(x.cuda())
    assert z_hip.shape == z_ref.shape and z_hip.is_cuda
    e = rel_err(z_hip, z_ref)
    assert e < 5e-4, f"logits rel err {e:.3e}"
    g = torch.randn_like(z_ref)
    (z_ref * g).sum().backward()
    (z_hip * g.cuda()).sum().backward()
    worst = ("", 0.0)
    ref_grads = {n: p.grad for n, p in ref.named_parameters()}
    for (name, p_ref), (_, p_hip) in zip(ref.named_parameters(), hip.named_parameters()):
        assert p_hip.grad is not None, name
        scale = p_ref.grad.abs().max().item()
        err = (p_hip.grad.cpu() - p_ref.grad).abs().max().item()
        if feeds_norm(ref, name):
            # a bias in front of a norm layer has an analytically ZERO gradient (the norm subtracts the mean):
            # both sides hold only the rounding residue of sum(dy); bound it against the weight gradient
            wscale = ref_grads[name[:-len("bias")] + "weight"].abs().max().item()
            assert p_hip.grad.abs().max().item() <= 2e-3 * wscale + 1e-4, f"{name}: zero-gradient bias holds {p_hip.grad.abs().max().item():.3e}"
            assert scale <= 2e-3 * wscale + 1e-4, name
            continue
        r = err / (scale + 1e-6)
        if r > worst[1]:
            worst = (name, r)
        assert err <= 2e-3 * scale + 2e-6, f"{name}: grad err {err:.3e} vs scale {scale:.3e}"
    print("worst grad", worst)
    # eval-mode forward (BatchNorm switches to running statistics on both sides)
    ref.eval()
    hip.eval()
    with torch.no_grad():
        e = rel_err(hip(x.cuda()), ref(x))
    assert e < 5e-4, f"eval logits rel err {e:.3e}"
    if "BATCH" in str(cfg["norm"]):
        for (k, b_ref), (_, b_hip) in zip(ref.named_buffers(), hip.named_buffers()):
            assert rel_err(b_hip.float(), b_ref.float()) < 1e-4, k


def test_unet_rejects_cpu_and_bad_extent():
    from multimodal_tta_amd.models import UNet
    from multimodal_tta_amd.ops import MmttaError
    m = UNet(SMALL)
    with pytest.raises(MmttaError):
        m(torch.randn(1, 4, 16, 16, 16))
    m = m.cuda()
    with pytest.raises(ValueError):
        m(torch.randn(1, 4, 16, 16, 24).cuda())


def test_torch_optimizer_step_through_the_facade():
    """reference SegTrainer.run_step skeleton (src/core/trainers/seg_trainer.py:105-145) with a stock
    torch optimizer: parameters move exactly as the oracle's do."""
    ref, hip = build_pair(SMALL)
    x = torch.randn(1, 4, 32, 32, 32)
    y = (torch.rand(1, 3, 32, 32, 32) > 0.5).float()
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    o_hip = torch.optim.Adam(hip.parameters(), lr=1e-3)
    for _ in range(2):
        for m, o, dev in ((ref, o_ref, "cpu"), (hip, o_hip, "cuda")):
            o.zero_grad()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(m(x.to(dev)), y.to(dev))
            loss.backward()
            o.step()
    with torch.no_grad():
        e = rel_err(hip(x.cuda()), ref(x))
    assert e < 2e-2, f"logits after 2 Adam steps rel err {e:.3e}"
