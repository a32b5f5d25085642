"""Seeded synthetic BraTS / HECKTOR-shaped volumes (no patient data ships with this repo).

Emits exactly the tensor contract of the reference datasets (src/datasets/brats.py:343-347,
394-401; src/datasets/hecktor21.py:290-298): ``image`` float32 [C,D,H,W] with background exactly
0 (brats.py:7), ``label`` float32 [R,D,H,W] in {0,1}, ``domain`` str, ``case_id`` str, ``index``.
Recipe: SURVEY.md section 8(d) - volume i uses ``torch.Generator().manual_seed(seed + i)``;
image = randn x centred-ellipsoid brain mask (semi-axes 0.45 * extent) plus a mild per-region
intensity bump so the regions are learnable; labels are nested ellipsoids
ET in TC in WT (semi-axes 0.10 / 0.16 / 0.24 * min extent) at a seeded offset.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch


def _ellipsoid(shape: Sequence[int], center, radii) -> torch.Tensor:
    D, H, W = shape
    z = torch.arange(D, dtype=torch.float32).view(D, 1, 1)
    y = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    x = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    q = ((z - center[0]) / radii[0]) ** 2 + ((y - center[1]) / radii[1]) ** 2 + ((x - center[2]) / radii[2]) ** 2
    return (q <= 1.0).float()


def synth_volume(index: int, channels: int, shape: Tuple[int, int, int], regions: int, seed: int = 42,
                 domain: str = "synth") -> Dict[str, object]:
    g = torch.Generator().manual_seed(int(seed) + int(index))
    D, H, W = shape
    img = torch.randn((channels, D, H, W), generator=g, dtype=torch.float32)
    brain = _ellipsoid(shape, ((D - 1) / 2, (H - 1) / 2, (W - 1) / 2), (0.45 * D, 0.45 * H, 0.45 * W))
    off = (torch.rand(3, generator=g) - 0.5) * 0.3
    center = ((D - 1) / 2 + off[0].item() * D, (H - 1) / 2 + off[1].item() * H, (W - 1) / 2 + off[2].item() * W)
    m = float(min(D, H, W))
    fracs = (0.10, 0.16, 0.24) if regions == 3 else tuple(0.16 + 0.04 * k for k in range(regions))
    masks = [_ellipsoid(shape, center, (f * m, f * m, f * m)) * brain for f in fracs]
    label = torch.stack(masks, dim=0)
    bump = torch.zeros((D, H, W))
    for k, mk in enumerate(masks):
        bump = bump + (0.6 + 0.2 * k) * mk
    img = (img + bump.unsqueeze(0)) * brain.unsqueeze(0)
    return {"image": img.contiguous(), "label": label.contiguous(), "domain": domain,
            "case_id": f"synth_{index:05d}", "index": int(index)}


class SyntheticSegDataset(torch.utils.data.Dataset):
    def __init__(self, num_volumes: int, channels: int, shape, regions: int, seed: int = 42, domain: str = "synth",
                 first_index: int = 0):
        self.n, self.c, self.shape, self.r = int(num_volumes), int(channels), tuple(int(s) for s in shape), int(regions)
        self.seed, self.domain, self.first = int(seed), str(domain), int(first_index)

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        return synth_volume(self.first + i, self.c, self.shape, self.r, self.seed, self.domain)
