"""HD95 / ASD of the evaluation tail (``mmtta_surface_distances``) against the scipy/torch restatement of MONAI's
algorithm in oracle/surface.py (reference src/evaluation/seg_eval.py:312-360).  PARITY UNPINNED: MONAI itself is not
installed, the oracle restates its published algorithm.

Tolerance: distances are float32(sqrt(exact squared distance)) on both sides; the GPU reproduces the float32
quantile formula of torch, so HD agrees to 1e-6 relative (one float32 ulp of an fma contraction at most); the GPU
mean is an exact fixed-point sum, torch's a float32 pairwise sum: ASD agrees to 2e-6 relative.  Empty-set cases must
match class for class (NaN / +inf).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def blobs(seed, shape, n=3, rmax=9):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    z, y, x = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    m = torch.zeros(shape, dtype=torch.bool)
    for _ in range(n):
        c = [float(torch.rand((), generator=g)) * s for s in shape]
        r = [2.0 + float(torch.rand((), generator=g)) * rmax for _ in range(3)]
        m |= (((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2) <= 1.0
    return m


def run(pred, gt, spacing=(1.0, 1.0, 1.0), percentile=95.0, symmetric=False):
    from multimodal_tta_amd import ops
    hd, asd = ops.surface_distances(pred.to(torch.uint8).cuda().contiguous(), gt.float().cuda(), spacing, percentile, symmetric)
    return hd.cpu(), asd.cpu()


def same(name, got, want, rel):
    if math.isnan(want):
        assert math.isnan(got), f"{name}: got {got}, oracle nan"
    elif math.isinf(want):
        assert got == want, f"{name}: got {got}, oracle {want}"
    else:
        assert abs(got - want) <= rel * abs(want) + 1e-7, f"{name}: got {got!r}, oracle {want!r}"


@pytest.mark.parametrize("shape,spacing,symmetric", [
    ((24, 40, 36), (1.0, 1.0, 1.0), False),
    ((17, 33, 70), (2.5, 0.9, 1.2), True),        # ragged extents (W > one 64-wide tile), anisotropic voxels
    ((48, 20, 9), (1.0, 3.0, 0.7), False),
])
def test_surface_distances_match_the_oracle(shape, spacing, symmetric):
    import oracle
    B, R = 2, 3
    pred = torch.stack([torch.stack([blobs(100 + 10 * b + r, shape) for r in range(R)]) for b in range(B)])
    gt = torch.stack([torch.stack([blobs(500 + 10 * b + r, shape) | (pred[b, r] & blobs(900 + r, shape, 2)) for r in range(R)])
                      for b in range(B)])
    gt[0, 1] = False                     # empty ground truth
    pred[1, 0] = False                   # empty prediction
    pred[1, 2] = False
    gt[1, 2] = False                     # both empty
    pred[0, 2, 0, :, :] |= gt[0, 2, 0, :, :]      # structures touching the volume border
    pred[0, 2, :, :, -1] = True
    hd, asd = run(pred, gt, spacing, 95.0, symmetric)
    for b in range(B):
        for r in range(R):
            hv, av = oracle.hd_asd(pred[b, r].numpy(), gt[b, r].numpy(), spacing, 95.0, symmetric)
            same(f"hd[{b},{r}]", float(hd[b, r]), hv, 1e-6)
            same(f"asd[{b},{r}]", float(asd[b, r]), av, 2e-6)


def test_percentiles_single_voxels_and_identity():
    import oracle
    shape = (12, 14, 16)
    a = torch.zeros(shape, dtype=torch.bool)
    b = torch.zeros(shape, dtype=torch.bool)
    a[3, 4, 5] = True
    b[9, 4, 1] = True                                # single voxels: distance sqrt(36 + 16)
    hd, asd = run(a[None, None], b[None, None], percentile=100.0)
    assert float(hd) == float(np.float32(math.sqrt(52.0))) and float(asd) == float(np.float32(math.sqrt(52.0)))
    m = blobs(7, shape, 2, 4)
    hd, asd = run(m[None, None], m[None, None])
    assert float(hd) == 0.0 and float(asd) == 0.0    # identical masks
    g = blobs(8, shape, 2, 4)
    for pct in (0.0, 50.0, 95.0, 100.0):
        hd, _ = run(m[None, None], g[None, None], percentile=pct)
        same(f"pct {pct}", float(hd), oracle.hd_asd(m.numpy(), g.numpy(), (1, 1, 1), pct)[0], 1e-6)
    from multimodal_tta_amd.ops import MmttaError
    with pytest.raises(MmttaError):
        run(m[None, None], g[None, None], percentile=101.0)
    hd1, asd1 = run(m[None, None], g[None, None])
    hd2, asd2 = run(m[None, None], g[None, None])
    assert torch.equal(hd1, hd2) and torch.equal(asd1, asd2)          # reproducible bit for bit


def test_full_volume_properties():
    """BASELINE-sized volume (128^3, R=3): too slow for scipy inside the GPU suite's budget at every region, so the
    size-independent properties are checked instead: symmetry of HD under swapping the masks, hd >= asd >= 0,
    translation of one mask by k voxels along W bounds |hd - hd0| <= k * spacing_w, and one region against scipy."""
    import oracle
    shape = (128, 128, 128)
    p = torch.stack([blobs(40 + r, shape, 4, 30) for r in range(3)])[None]
    g = torch.stack([blobs(60 + r, shape, 4, 30) for r in range(3)])[None]
    sp = (1.0, 1.0, 2.0)
    hd_pg, asd_pg = run(p, g, sp, 95.0, True)
    hd_gp, asd_gp = run(g, p, sp, 95.0, True)
    assert torch.equal(hd_pg, hd_gp) and torch.allclose(asd_pg, asd_gp, rtol=1e-6)
    assert bool((hd_pg >= 0).all()) and bool((asd_pg >= 0).all())
    shifted = torch.zeros_like(p)
    shifted[..., 3:] = p[..., :-3]
    hd_s, _ = run(shifted, p, sp)
    assert bool((hd_s <= 3 * sp[2] + 1e-5).all()), hd_s
    hv, av = oracle.hd_asd(p[0, 0].numpy(), g[0, 0].numpy(), sp, 95.0, True)
    same("hd 128^3", float(hd_pg[0, 0]), hv, 1e-6)
    same("asd 128^3", float(asd_pg[0, 0]), av, 2e-6)


def test_evaluator_reports_surface_metrics():
    """seg_eval with evaluation.surface.enable: keys and values follow reference seg_eval.py:342-355,424-440,459-476
    (penalty for an empty prediction, non-finite -> diagonal, regions with empty GT skipped)."""
    import oracle
    from test_hip_tta import SMALL, build_pair, root_cfg
    from multimodal_tta_amd.registry import get_dataset_builder, get_evaluation_strategy
    cfg = root_cfg(SMALL)
    cfg["dataset"]["synthetic"]["num_volumes"] = 2
    cfg["dataset"]["synthetic"]["shape"] = [32, 32, 32]
    cfg["evaluation"]["surface"] = {"enable": True, "asd_symmetric": True}
    cfg["evaluation"]["seg"]["spacing"] = [1.0, 1.5, 2.0]
    _, hip = build_pair(SMALL)
    loader = get_dataset_builder("brats")(cfg).get_loader("test")
    strat = get_evaluation_strategy("seg_eval")(cfg)
    got = strat.evaluate_epoch(hip, loader, torch.device("cuda"))
    regions = ["et", "tc", "wt"]
    sums = {k: [0.0, 0] for k in [f"{r}_{m}" for r in regions for m in ("hd95", "asd")]}
    with torch.no_grad():
        for batch in loader:
            z = hip(batch["image"].cuda()).cpu()
            pred, gt = oracle.masks_from_logits(z, batch["label"], 0.5)
            hd, asd = oracle.evaluator_surface(pred, gt, [1.0, 1.5, 2.0], True)
            valid = gt.reshape(*gt.shape[:2], -1).sum(-1) > 0
            for b in range(pred.shape[0]):
                for r, name in enumerate(regions):
                    if bool(valid[b, r]):
                        sums[f"{name}_hd95"][0] += float(hd[b, r]); sums[f"{name}_hd95"][1] += 1
                        sums[f"{name}_asd"][0] += float(asd[b, r]); sums[f"{name}_asd"][1] += 1
    for k, (s, n) in sums.items():
        want = s / n if n else 0.0
        assert abs(got[k] - want) <= 2e-6 * abs(want) + 1e-7, (k, got[k], want)
        assert f"dom/synth/{k}" in got
    assert {"avg_hd95", "avg_asd", "dom/synth/avg_hd95", "dom/synth/avg_asd"} <= set(got)
    # the adaptation evaluator carries the same columns through its gather table
    cfg["method"]["steps"] = 1
    m = get_evaluation_strategy("seg_tta_eval")(cfg).evaluate_epoch(hip, loader, torch.device("cuda"))
    assert {"avg_hd95", "avg_asd", "wt_hd95", "dom/synth/wt_asd"} <= set(m) and math.isfinite(m["avg_hd95"])
