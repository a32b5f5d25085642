"""NIfTI reader and the on-disk dataset builders (reference src/datasets/brats.py,
hecktor21.py).  PARITY UNPINNED: nibabel is not installed and the reference ships no NIfTI fixture, so the reader is
checked against files written here (known voxel patterns under known affines) and the builders against the
reference's documented selection rules.
"""
import os

import numpy as np
import pytest
import torch

from multimodal_tta_amd import nifti


def pattern(shape):
    x, y, z = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    return (x * 10000 + y * 100 + z).astype(np.float32)          # value encodes the voxel's own (i, j, k)


def test_identity_and_gzip_roundtrip(tmp_path):
    a = pattern((5, 6, 7))
    for name in ("a.nii", "a.nii.gz"):
        p = str(tmp_path / name)
        nifti.write_nifti(p, a, np.diag([1.0, 2.0, 3.0, 1.0]))
        got, aff = nifti.load(p)
        assert got.dtype == np.float32 and got.shape == (5, 6, 7) and np.array_equal(got, a)
        assert np.allclose(aff, np.diag([1.0, 2.0, 3.0, 1.0]))
        assert np.array_equal(nifti.load_canonical(p), a)         # already RAS+: untouched


def test_closest_canonical_flips_and_permutes(tmp_path):
    a = pattern((4, 5, 6))
    # LPS file: first two axes run towards left / posterior -> both are flipped
    p = str(tmp_path / "lps.nii.gz")
    nifti.write_nifti(p, a, np.diag([-1.0, -1.0, 1.0, 1.0]))
    assert np.array_equal(nifti.load_canonical(p), a[::-1, ::-1, :])
    # array axes (i, j, k) = world (z, x, -y): canonical array is indexed (x, y, z) = (j, reversed k, i)
    aff = np.zeros((4, 4))
    aff[2, 0], aff[0, 1], aff[1, 2], aff[3, 3] = 2.0, 1.0, -1.5, 1.0
    p = str(tmp_path / "perm.nii")
    nifti.write_nifti(p, a, aff)
    got = nifti.load_canonical(p)
    assert got.shape == (5, 6, 4)
    assert np.array_equal(got, a.transpose(1, 2, 0)[:, ::-1, :])
    # slightly oblique affine still snaps to the closest axes
    th = np.deg2rad(10.0)
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    aff = np.eye(4)
    aff[:3, :3] = rot @ np.diag([-1.0, 1.0, 1.0])
    p = str(tmp_path / "oblique.nii")
    nifti.write_nifti(p, a, aff)
    assert np.array_equal(nifti.load_canonical(p), a[::-1])
    assert np.array_equal(nifti.io_orientation(aff), [[0, -1], [1, 1], [2, 1]])


def test_qform_scaling_dtypes_and_endianness(tmp_path):
    a = (pattern((3, 4, 5)) % 120).astype(np.int16)
    # qform only: quaternion (b, c, d) = (0, 0, 1) is a 180-degree turn about z -> x and y flipped; qfac -1 flips z
    p = str(tmp_path / "q.nii")
    nifti.write_nifti(p, a, np.eye(4), use_qform=(0.0, 0.0, 1.0, -1.0))
    h = nifti.read_header(open(p, "rb").read())
    assert h["qform_code"] == 1 and h["sform_code"] == 0
    assert np.allclose(h["affine"][:3, :3], np.diag([-1.0, -1.0, -1.0]), atol=1e-6)
    assert np.array_equal(nifti.load_canonical(p), a[::-1, ::-1, ::-1].astype(np.float32))
    # scl_slope / scl_inter, big-endian payload
    p = str(tmp_path / "be.nii.gz")
    nifti.write_nifti(p, a, np.eye(4), slope=0.5, inter=-3.0, endian=">")
    got, _ = nifti.load(p)
    assert got.dtype == np.float32 and np.array_equal(got, a.astype(np.float32) * 0.5 - 3.0)
    # slope 0 means "no scaling"
    p = str(tmp_path / "s0.nii")
    nifti.write_nifti(p, a, np.eye(4), slope=0.0, inter=7.0)
    assert np.array_equal(nifti.load(p)[0], a.astype(np.float32))
    for dt in (np.uint8, np.int32, np.float64, np.uint16):
        p = str(tmp_path / f"{np.dtype(dt).name}.nii")
        nifti.write_nifti(p, a.astype(dt), np.eye(4))
        assert np.array_equal(nifti.load(p)[0], a.astype(np.float32))
    # neither sform nor qform: base affine, first axis right -> left, so the canonical array is x-flipped
    p = str(tmp_path / "none.nii")
    nifti.write_nifti(p, a, np.eye(4))
    raw = bytearray(open(p, "rb").read())
    raw[252:256] = b"\x00\x00\x00\x00"
    open(p, "wb").write(bytes(raw))
    assert np.array_equal(nifti.load_canonical(p), a[::-1].astype(np.float32))


def test_malformed_files_are_refused(tmp_path):
    p = str(tmp_path / "short.nii")
    open(p, "wb").write(b"\x00" * 100)
    with pytest.raises(nifti.NiftiError, match="shorter"):
        nifti.load(p)
    a = pattern((3, 3, 3))
    p = str(tmp_path / "trunc.nii")
    nifti.write_nifti(p, a)
    whole = open(p, "rb").read()
    open(p, "wb").write(whole[:-8])
    with pytest.raises(nifti.NiftiError, match="truncated"):
        nifti.load(p)
    raw = bytearray(whole)
    raw[344:348] = b"ni1\x00"
    open(p, "wb").write(bytes(raw))
    with pytest.raises(nifti.NiftiError, match="two-file"):
        nifti.load(p)
    raw[0:4] = (540).to_bytes(4, "little")
    open(p, "wb").write(bytes(raw))
    with pytest.raises(nifti.NiftiError, match="NIfTI-2"):
        nifti.load(p)


# ----------------------------------------------------------------------------- builders
def brats_tree(root, n_cases=5, shape=(6, 8, 10), profile_rows=True):
    import pandas as pd
    rows = []
    rng = np.random.RandomState(0)
    for i in range(n_cases):
        cid = f"case{i:03d}"
        split = ["train", "train", "val", "test", "test"][i % 5]
        lab = rng.randint(0, 5, size=shape).astype(np.float32)
        lp = os.path.join(root, f"{cid}_seg.nii.gz")
        nifti.write_nifti(lp, lab, np.diag([-1.0, -1.0, 1.0, 1.0]))
        mods = ["t1n", "t1c", "t2w", "t2f"] if i != 1 else ["t1n", "t1c", "t2w"]      # case001 lacks t2f -> dropped
        for m, mod in enumerate(mods):
            ip = os.path.join(root, f"{cid}_{mod}.nii.gz")
            nifti.write_nifti(ip, pattern(shape) + 1000000 * (m + 1) + i, np.diag([-1.0, -1.0, 1.0, 1.0]))
            rows.append({"subject_id": cid, "modality": mod.upper(), "img_path": os.path.basename(ip),
                         "label_path": os.path.basename(lp) if i != 4 else np.nan, "split": split})
    csv = os.path.join(root, "processed.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    return csv


def brats_cfg(csv, root, **src):
    from multimodal_tta_amd.config import compose
    cfg = compose(overrides=["task=brats", "model=unet"])
    cfg["dataset"]["synthetic"]["enabled"] = False
    cfg["dataset"]["expected_shape"] = [6, 8, 10]
    s = {"name": "ssa_site", "profile": "ped", "csv_path": csv, "root_dir": root,
         "include_splits": {"train": [], "val": [], "test": ["train", "test"]}}
    s.update(src)
    cfg["dataset"]["sources"] = [s]
    cfg["training"]["num_workers"] = 0
    cfg["training"]["eval_batch_size"] = 1
    cfg["training"]["data"]["transforms"]["image_size"] = [10, 8, 6]
    return cfg


def test_brats_builder_follows_the_reference_rules(tmp_path):
    from multimodal_tta_amd.registry import get_dataset_builder
    root = str(tmp_path)
    csv = brats_tree(root)
    b = get_dataset_builder("brats")(brats_cfg(csv, root))
    assert b.get_dataset("val") is None and b.get_loader("train") is None        # disabled for every source
    ds = b.get_dataset("test")
    # case001 lacks a modality, case004 has no label (drop_unlabeled); 'test' pulls the csv's train + test rows
    assert [ds[i]["case_id"] for i in range(len(ds))] == ["case000", "case003"]
    item = ds[0]
    assert item["image"].shape == (4, 10, 8, 6) and item["image"].dtype == torch.float32      # (C, Z, Y, X)
    assert item["label"].shape == (3, 10, 8, 6) and item["domain"] == "ssa_site" and item["profile"] == "ped"
    # the LPS files are flipped to RAS+, then (X,Y,Z) -> (Z,Y,X): voxel (z,y,x) holds the pattern of (5-x, 7-y, z)
    z, y, x = 3, 2, 1
    assert float(item["image"][0, z, y, x]) == (5 - x) * 10000 + (7 - y) * 100 + z + 1000000
    assert float(item["image"][3, z, y, x]) == (5 - x) * 10000 + (7 - y) * 100 + z + 4000000
    raw = nifti.load_canonical(os.path.join(root, "case000_seg.nii.gz")).transpose(2, 1, 0)
    for r, ids in enumerate(([1], [1, 2, 3], [1, 2, 3, 4])):                     # PED taxonomy (brats.py:70-75)
        assert torch.equal(item["label"][r], torch.from_numpy(np.isin(raw, ids).astype(np.float32)))
    batch = next(iter(b.get_loader("test")))
    assert batch["image"].shape == (1, 4, 10, 8, 6) and batch["domain"] == ["ssa_site"] and int(batch["index"][0]) == 0
    shard = b.get_dataset("test", shard=(1, 2))
    assert len(shard) == 1 and shard[0]["case_id"] == "case003"
    # region_map override and a wrong on-disk shape
    b2 = get_dataset_builder("brats")(brats_cfg(csv, root, region_map={"ET": [4], "TC": [4], "WT": [1, 4]}))
    assert torch.equal(b2.get_dataset("test")[0]["label"][2], torch.from_numpy(np.isin(raw, [1, 4]).astype(np.float32)))
    bad = brats_cfg(csv, root)
    bad["dataset"]["expected_shape"] = [6, 8, 11]
    with pytest.raises(ValueError, match="Shape mismatch"):
        get_dataset_builder("brats")(bad).get_dataset("test")[0]
    bad = brats_cfg(csv, root)
    bad["training"]["data"]["transforms"]["image_size"] = [10, 8, 7]
    with pytest.raises(ValueError, match="spatial mismatch"):
        get_dataset_builder("brats")(bad).get_dataset("test")[0]
    none = brats_cfg(csv, root, include_splits={"train": [], "val": [], "test": ["nothing"]})
    with pytest.raises(ValueError, match="No samples after filtering"):
        get_dataset_builder("brats")(none).get_dataset("test")
    missing = brats_cfg(os.path.join(root, "absent.csv"), root)
    with pytest.raises(FileNotFoundError):
        get_dataset_builder("brats")(missing).get_dataset("test")


def hecktor_tree(root, shape=(6, 6, 4)):
    import pandas as pd
    rows = []
    centers = ["CHUM"] * 4 + ["CHGJ"] * 3 + ["chus"] * 3
    for i, c in enumerate(centers):
        pid = f"P{i:02d}"
        for kind, scale in (("ct", 1.0), ("pt", 2.0)):
            nifti.write_nifti(os.path.join(root, f"{pid}_{kind}.nii.gz"), pattern(shape) * scale + i)
        lab = np.zeros(shape, np.uint8)
        lab[1:3, 2:4, 1:3] = 255 if i % 2 else 1
        nifti.write_nifti(os.path.join(root, f"{pid}_gtvt.nii.gz"), lab)
        rows.append({"patient_id": pid, "status": "ok" if i != 2 else "failed", "ct_proc": f"{pid}_ct.nii.gz",
                     "pt_proc": f"{pid}_pt.nii.gz", "gtvt_proc": f"{pid}_gtvt.nii.gz", "center_code": c,
                     "center_id": i % 3})
    csv = os.path.join(root, "manifest.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    return csv


def test_hecktor_builder_splits_around_the_target_centre(tmp_path):
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.datasets import sample_val_indices_per_center
    from multimodal_tta_amd.registry import get_dataset_builder
    root = str(tmp_path)
    csv = hecktor_tree(root)
    cfg = compose(overrides=["task=hecktor21", "model=unet"])
    cfg["dataset"]["synthetic"]["enabled"] = False
    cfg["dataset"].update({"manifest_csv": csv, "root_dir": root, "expected_shape": [6, 6, 4], "target_center": "chus",
                           "val_per_center": 1, "split_seed": 2026})
    cfg["training"]["num_workers"] = 0
    cfg["training"]["data"]["transforms"]["image_size"] = [4, 6, 6]
    cfg["training"]["data"]["transforms"]["geom_aug"] = False
    cfg["training"]["data"]["transforms"]["intensity_aug"] = False
    b = get_dataset_builder("hecktor21")(cfg)
    test, val, train = b.get_dataset("test"), b.get_dataset("val"), b.get_dataset("train")
    assert [test[i]["case_id"] for i in range(len(test))] == ["P07", "P08", "P09"]       # target centre, upper-cased
    assert all(test[i]["domain"] == "CHUS" for i in range(len(test)))
    # one validation case per remaining centre, drawn by RandomState(2026) over the centres in sorted order; P02 failed QC
    want = sample_val_indices_per_center({"CHGJ": np.array([4, 5, 6]), "CHUM": np.array([0, 1, 3])}, 1, 2026)
    assert sorted(val[i]["case_id"] for i in range(len(val))) == sorted(f"P{k:02d}" for k in want)
    got_train = sorted(train[i]["case_id"] for i in range(len(train)))
    assert got_train == sorted(f"P{k:02d}" for k in (0, 1, 3, 4, 5, 6) if k not in want)
    item = test[1]
    assert item["image"].shape == (2, 4, 6, 6) and item["label"].shape == (1, 4, 6, 6) and item["center_id"] == 8 % 3
    assert set(torch.unique(item["label"]).tolist()) == {0.0, 1.0}                        # {0,255} mapped to {0,1}
    assert float(item["label"].sum()) == 8.0 and float(item["label"][0, 1, 2, 1]) == 1.0  # (z,y,x) = (1,2,1)
    assert float(item["image"][1, 3, 2, 1]) == 2.0 * (1 * 10000 + 2 * 100 + 3) + 8
    cfg["dataset"]["target_center"] = "NOPE"
    with pytest.raises(ValueError, match="0 samples"):
        get_dataset_builder("hecktor21")(cfg).get_dataset("test")
    cfg["dataset"]["target_center"] = "chus"
    cfg["training"]["data"]["transforms"]["geom_aug"] = True
    with pytest.raises(NotImplementedError):
        get_dataset_builder("hecktor21")(cfg).get_dataset("train")
    assert len(get_dataset_builder("hecktor21")(cfg).get_dataset("test")) == 3           # eval splits never augment


def test_evaluator_prepass_default_follows_the_data_source():
    from multimodal_tta_amd.evaluation import SegmentationEvaluationStrategy
    assert not SegmentationEvaluationStrategy({"dataset": {"synthetic": {"enabled": True}}}).normalize_on_device
    assert SegmentationEvaluationStrategy({"dataset": {"synthetic": {"enabled": False}}}).normalize_on_device
    off = {"dataset": {"synthetic": {"enabled": False}}, "training": {"data": {"transforms": {"normalize_on_device": False}}}}
    assert not SegmentationEvaluationStrategy(off).normalize_on_device
