"""Sharding and the gather/merge of per-volume results: N simulated ranks == one rank, and a real
2-process gloo run of the single collective of the path."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from multimodal_tta_amd.evaluation import (RegionAccumulator, gather_table, metrics_from_table, shard_indices,
                                           table_width)

R = 3
REGIONS = ["ET", "TC", "WT"]


def fake_rows(n, surface=False):
    g = torch.Generator().manual_seed(5)
    rows = []
    for i in range(n):
        dice, iou = torch.rand(R, generator=g, dtype=torch.float64), torch.rand(R, generator=g, dtype=torch.float64)
        valid = (torch.rand(R, generator=g) > 0.3).double()
        parts = [torch.tensor([i, i % 2, 0.1 * i], dtype=torch.float64), dice.float().double(), iou.float().double(), valid]
        if surface:      # hd95[R], asd[R] (evaluation.surface.enable)
            parts += [(40.0 * torch.rand(R, generator=g)).double(), (9.0 * torch.rand(R, generator=g)).double()]
        rows.append(torch.cat(parts))
    return torch.stack(rows)


@pytest.mark.parametrize("n,world", [(7, 2), (8, 8), (5, 8), (64, 8), (1, 4)])
def test_shards_partition_and_merge_equals_single_rank(n, world):
    shards = [shard_indices(n, r, world) for r in range(world)]
    assert sorted(i for s in shards for i in s) == list(range(n))
    assert max(len(s) for s in shards) <= (n + world - 1) // world
    rows = fake_rows(n)
    want = metrics_from_table(rows, REGIONS, ["a", "b"], True)
    # simulate the all_gather: pad each shard to ceil(N/W), concatenate in rank order, drop pads, sort
    per = (n + world - 1) // world
    bufs = []
    for s in shards:
        pad = torch.full((per, table_width(R)), -1.0, dtype=torch.float64)
        if s:
            pad[:len(s)] = rows[s]
        bufs.append(pad)
    allrows = torch.cat(bufs)
    allrows = allrows[allrows[:, 0] >= 0]
    merged = allrows[torch.argsort(allrows[:, 0])]
    assert torch.equal(merged, rows)
    assert metrics_from_table(merged, REGIONS, ["a", "b"], True) == want


def test_metrics_from_table_equals_accumulator():
    rows = fake_rows(6)
    acc = RegionAccumulator(REGIONS)
    for row in rows:
        acc.add_row(row[3:6].float().tolist(), row[6:9].float().tolist(), (row[9:12] > 0.5).tolist(), ["a", "b"][int(row[1])])
        acc.add_loss(float(row[2]), 1)
    assert acc.metrics(True) == metrics_from_table(rows, REGIONS, ["a", "b"], True)


def _worker(rank, world, port, n, out_dir, surface=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = fake_rows(n, surface)
    mine = rows[shard_indices(n, rank, world)]
    table = gather_table(mine, n, world)
    torch.save(table, os.path.join(out_dir, f"t{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("surface", [False, True])
def test_gather_table_two_processes_gloo(tmp_path, surface):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 5, 2
    mp.spawn(_worker, args=(world, port, n, str(tmp_path), surface), nprocs=world, join=True)
    want = fake_rows(n, surface)
    assert want.shape[1] == table_width(R, surface)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"t{r}.pt"), weights_only=True)
        assert torch.equal(got, want), f"rank {r}"
    if surface:      # every rank replays the same aggregation, surface keys included
        m = metrics_from_table(want, REGIONS, ["a", "b"], True, surface=True)
        assert "avg_hd95" in m and "dom/a/avg_asd" in m and m["avg_hd95"] > 0.0


# ----------------------------------------------------------------------------- seg_eval / merge under a real group
class _CpuScoredEval:
    """Mixin for the CPU tests: the strategy's host logic is the product code, only the voxel kernel behind
    ``score`` (mmtta_mask_dice_counts, GPU-only) is replaced by the oracle's restatement of the same counts."""

    def score(self, logits, y, channels_last=False):
        import oracle
        pred, gt = oracle.masks_from_logits(logits, y, self.threshold)
        inter = (pred & gt).flatten(2).sum(-1)
        return torch.stack([inter, pred.flatten(2).sum(-1), gt.flatten(2).sum(-1)], dim=-1).to(torch.int64)


def _eval_volumes(n):
    g = torch.Generator().manual_seed(11)
    vols = []
    for i in range(n):
        x = torch.randn(1, 2, 4, 6, 6, generator=g)
        y = (torch.rand(1, R, 4, 6, 6, generator=g) > 0.6).float()
        if i == 2:
            y[:, 1] = 0.0          # an empty ground-truth region: valid = False on that row
        vols.append((x, y, ["siteA", "siteB", "siteC"][i % 3]))
    return vols


def _eval_setup():
    from multimodal_tta_amd.evaluation import SegmentationEvaluationStrategy

    class Strat(_CpuScoredEval, SegmentationEvaluationStrategy):
        pass

    cfg = {"evaluation": {"seg": {"threshold": 0.5, "region_order": REGIONS}, "loss": {"report_loss": False}},
           "dataset": {"synthetic": {"enabled": True}}}
    torch.manual_seed(3)
    model = torch.nn.Conv3d(2, R, 1)
    return Strat(cfg), model


def _batches(vols, indices, bs):
    out = []
    for j in range(0, len(indices), bs):
        ids = indices[j:j + bs]
        out.append({"image": torch.cat([vols[i][0] for i in ids]), "label": torch.cat([vols[i][1] for i in ids]),
                    "domain": [vols[i][2] for i in ids], "index": torch.tensor(ids)})
    return out


def _seg_eval_worker(rank, world, port, n, out_dir, shards):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    strat, model = _eval_setup()
    metrics = strat.evaluate_epoch(model, _batches(_eval_volumes(n), shards[rank], 2), "cpu")
    with open(os.path.join(out_dir, f"m{rank}.json"), "w") as fh:
        json.dump(metrics, fh)
    torch.save(strat.last_table, os.path.join(out_dir, f"tab{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("shards", [[[0, 2, 4], [1, 3]], [[0, 1, 2, 3], [4]], [[0, 1, 2, 3, 4], []]])
def test_seg_eval_under_two_gloo_ranks_reports_the_whole_split(tmp_path, shards):
    """ADVICE r1 (main.py:57): a plain ``seg_eval`` run under torch.distributed must merge the shards - every rank
    returns the metrics of all volumes, identical to one process; unequal (and empty) shards included."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 5, 2
    strat, model = _eval_setup()
    want = strat.evaluate_epoch(model, _batches(_eval_volumes(n), list(range(n)), 2), "cpu")
    assert "dom/siteA/avg_dc" in want and want["avg_dc"] > 0.0
    mp.spawn(_seg_eval_worker, args=(world, port, n, str(tmp_path), shards), nprocs=world, join=True)
    tabs = []
    for r in range(world):
        with open(os.path.join(str(tmp_path), f"m{r}.json")) as fh:
            got = json.load(fh)
        assert got == want, f"rank {r}: {got} vs {want}"
        tabs.append(torch.load(os.path.join(str(tmp_path), f"tab{r}.pt"), weights_only=True))
    assert torch.equal(tabs[0], tabs[1]) and tabs[0].shape == (n, table_width(R))
    assert tabs[0][:, 0].tolist() == [float(i) for i in range(n)]


# ----------------------------------------------------------------------------- bench.py's table assembly
def _bench_counts(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    gt = torch.randint(0, 5000, (n, R), generator=g)
    gt[0, 1] = 0                                   # an empty ground-truth region: valid = False
    pred = torch.randint(0, 5000, (n, R), generator=g)
    inter = torch.minimum(gt, pred) // 2
    return torch.stack([inter, pred, gt], dim=-1).to(torch.int64)


def _bench_worker(rank, world, port, n, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = bench.bench_rows(_bench_counts(rank, n), rank, world)
    table = gather_table(rows, n * world, world)
    torch.save(table, os.path.join(out_dir, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_bench_table_assembly_two_processes_gloo(tmp_path):
    """bench.py --gpus N: every rank contributes `steps` rows with global indices i*W + rank; the gathered table is the
    same on every rank, sorted by volume index, and post_tta_dice is the mean over the VALID (volume, region) pairs."""
    import bench
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 3, 2
    mp.spawn(_bench_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    per_rank = [bench.bench_rows(_bench_counts(r, n), r, world) for r in range(world)]
    want = torch.cat(per_rank)
    want = want[torch.argsort(want[:, 0])]
    assert want[:, 0].tolist() == [float(i) for i in range(n * world)]
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"b{r}.pt"), weights_only=True)
        assert torch.equal(got, want), f"rank {r}"
    valid = want[:, 3 + 2 * R:3 + 3 * R] > 0.5
    assert int((~valid).sum()) == world              # the two empty-GT regions are excluded
    assert abs(bench.post_dice_from_table(want, R) - float(want[:, 3:3 + R][valid].mean())) < 1e-12


# ----------------------------------------------------------------------------- optional mask gather (north star: "Dice/logits")
def _mask_of(i):
    g = torch.Generator().manual_seed(1000 + i)
    shape = (R, 4 + i % 2, 6, 5 + i % 3)                 # ragged extents across volumes (HECKTOR-style crops)
    return (torch.rand(shape, generator=g) > 0.5).to(torch.uint8)


def _mask_worker(rank, world, port, shards, out_dir):
    import torch.distributed as dist
    from multimodal_tta_amd.evaluation import gather_masks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = gather_masks([(i, _mask_of(i)) for i in shards[rank]], "cpu")
    torch.save({int(k): v for k, v in got.items()}, os.path.join(out_dir, f"mk{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("shards", [[[0, 2, 4], [1, 3]], [[0, 1, 2, 3, 4], []]])
def test_gather_masks_two_processes_gloo(tmp_path, shards):
    """`evaluation.gather_masks`: the second collective of a sharded evaluation - every rank ends up with the uint8 mask of
    EVERY volume, equal to the single-process masks; unequal and empty shards, ragged extents."""
    from multimodal_tta_amd.evaluation import gather_masks
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    single = gather_masks([(i, _mask_of(i)) for i in range(5)], "cpu")
    assert sorted(single) == list(range(5))
    mp.spawn(_mask_worker, args=(2, port, shards, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"mk{r}.pt"), weights_only=True)
        assert sorted(got) == list(range(5)), f"rank {r}"
        for i in range(5):
            assert got[i].dtype == torch.uint8 and torch.equal(got[i], single[i]), f"rank {r}, volume {i}"


def test_gather_table_drops_repeated_volumes():
    """A sampler that pads the last shard repeats volumes: the merged table keeps one row per index."""
    rows = fake_rows(4)
    dup = torch.cat([rows, rows[1:2]])
    assert torch.equal(gather_table(dup, 5, 1), rows)


# ----------------------------------------------------------------------------- 8 ranks (the first real 8-GPU run must need no debugging)
def _bench8_worker(rank, world, port, n_total, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_total, rank, world)           # N = 5: ranks 5..7 hold nothing
    counts = _bench_counts(rank, max(len(mine), 1))[:len(mine)]
    rows = bench.bench_rows(counts, rank, world) if mine else torch.empty((0, table_width(R)), dtype=torch.float64)
    table = gather_table(rows, n_total, world)
    torch.save(table, os.path.join(out_dir, f"e{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 64])
def test_eight_rank_table_assembly_gloo(tmp_path, n_total):
    """BASELINE configs 4 / 5 shard the test split over 8 ranks: N = 5 (empty ranks) and N = 64 (8 volumes per rank = 2 lanes
    x a group of 4) produce the same index-sorted table on every rank, equal to the single-process assembly."""
    import bench
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 8
    mp.spawn(_bench8_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    parts = []
    for r in range(world):
        mine = shard_indices(n_total, r, world)
        if mine:
            parts.append(bench.bench_rows(_bench_counts(r, len(mine)), r, world))
    want = torch.cat(parts)
    want = want[torch.argsort(want[:, 0])]
    assert want[:, 0].tolist() == [float(i) for i in range(n_total)]
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"e{r}.pt"), weights_only=True)
        assert torch.equal(got, want), f"rank {r}"
