"""Multi-rank host logic on CPU: band partition, view split and the band gather over gloo with
two and three processes.  The renderer in these processes is the CPU oracle (test infrastructure) standing
in for the GPU, because the point here is the split/gather code, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from renderer_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_rows_partition_the_frame():
    for h in (600, 2160, 4320, 37, 1080):
        for world in (1, 2, 3, 4, 8):
            bands = shard.all_bands(h, world)
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                assert a1 == b0 and a0 <= a1
            for y0, y1 in bands:
                if y1 > y0:  # ranks beyond the last tile row get an empty band
                    assert y0 % shard.TILE == 0 and (y1 % shard.TILE == 0 or y1 == h)
            rows = [(y1 - y0 + shard.TILE - 1) // shard.TILE for y0, y1 in bands]
            assert max(rows) - min(rows) <= 1
    with pytest.raises(ValueError):
        shard.band_rows(100, 2, 2)


def test_weighted_bands_and_the_balance_against_the_gather():
    """Band heights in given proportions still partition the frame into whole tile rows; the balance
    formula gives rank 0 the taller band exactly when the senders' links are the slower side."""
    rng = np.random.default_rng(4)
    for h in (2160, 4320, 1080, 100):
        for world in (2, 3, 4, 8):
            for _ in range(20):
                w = rng.uniform(0.05, 1.0, world).tolist()
                bands = shard.all_bands(h, world, weights=w)
                assert bands[0][0] == 0 and bands[-1][1] == h
                for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                    assert a1 == b0 and a0 <= a1
                for y0, y1 in bands:
                    if y1 > y0:
                        assert y0 % shard.TILE == 0 and (y1 % shard.TILE == 0 or y1 == h)
                n_rows = (h + shard.TILE - 1) // shard.TILE
                if n_rows >= 4 * world:  # heights follow the weights to within a tile row or two
                    for (y0, y1), x in zip(bands, w):
                        assert abs((y1 - y0) / shard.TILE - n_rows * x / sum(w)) < 2.0
                    assert all(y1 > y0 for y0, y1 in bands)
    assert shard.all_bands(2160, 4, weights=[1, 1, 1, 1]) == shard.all_bands(2160, 4)
    assert shard.all_bands(2160, 3, weights=[0, 1, 1])[0] == (0, 0)
    for bad in ([1, 1], [0, 0, 0], [1, 1, 1, 1]):
        with pytest.raises(ValueError):
            shard.band_rows(2160, 3, 0, weights=bad)
    # two ranks, a frame costs 0.075 ms apart from its rows, 0.143 ms for all rows, 0.553 ms over one link:
    # equal bands would wait 0.277 ms for half a frame; rank 0 takes the share s with 0.075 + 0.143 s = 0.553 (1 - s)
    w = shard.balanced_weights(2, 0, 0.075, 0.143, 0.553)
    s0 = (0.553 - 0.075) / (0.143 + 0.553)
    assert abs(w[0] - s0) < 1e-12 and abs(sum(w) - 1.0) < 1e-12 and abs((0.075 + 0.143 * w[0]) - 0.553 * w[1]) < 1e-12
    w = shard.balanced_weights(4, 0, 0.075, 0.143, 0.553)
    assert w[0] > w[1] == w[2] == w[3] and abs((0.075 + 0.143 * w[0]) - 0.553 * w[1]) < 1e-12
    assert shard.balanced_weights(4, 2, 0.075, 0.143, 0.553)[2] == w[0]
    # eight ranks: an eighth of a frame is sent faster than it is drawn, equal bands stay
    assert shard.balanced_weights(8, 0, 0.075, 0.143, 0.553) == [1.0] * 8
    assert shard.balanced_weights(1, 0, 0.1, 0.1, 9.0) == [1.0]
    assert shard.balanced_weights(2, 0, 0.1, 0.1, float("nan")) == [1.0, 1.0]
    assert shard.balanced_weights(2, 0, 0.0, 0.001, 100.0)[0] == 0.9  # (never more than nine tenths)


def test_views_for_rank_cover_all_views():
    for n in (4096, 7, 1):
        for world in (1, 2, 8):
            got = [shard.views_for_rank(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from renderer_amd import scenes, shard as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = scenes.config_scene(2, scale=0.1)
    w, h = s.size
    models = oracle.to_model(s.transforms)
    view, proj = oracle.camera(s.camera, w, h)
    y0, y1 = sh.band_rows(h, world, rank)
    heights = [b - a for a, b in sh.all_bands(h, world)]
    if world == 3:
        assert len(set(heights)) > 1, "this case is meant to have bands of different heights"
    band = oracle.render(s, models, view, proj, w, h, scissor=(y0, y1))["color"][y0:y1]
    frame = sh.gather_bands(torch.from_numpy(np.ascontiguousarray(band)), h, w, world, rank)
    # a second frame into a caller-provided buffer, as the overlapped pipeline does it
    out = torch.full((h, w, 4), 7, dtype=torch.uint8) if rank == 0 else None
    again = sh.gather_bands(torch.from_numpy(np.ascontiguousarray(band)), h, w, world, rank, out=out)
    # bands in given proportions (rank 0, which sends nothing, takes the tallest)
    wts = sh.balanced_weights(world, 0, 0.075, 0.143, 0.553)
    b0, b1 = sh.band_rows(h, world, rank, weights=wts)
    wband = oracle.render(s, models, view, proj, w, h, scissor=(b0, b1))["color"][b0:b1]
    weighted = sh.gather_bands(torch.from_numpy(np.ascontiguousarray(wband)), h, w, world, rank, weights=wts)
    # the whole frame on every rank, equal and weighted bands
    everywhere = sh.exchange_bands(torch.from_numpy(np.ascontiguousarray(band)), h, w, world, rank)
    everywhere_w = sh.exchange_bands(torch.from_numpy(np.ascontiguousarray(wband)), h, w, world, rank, weights=wts,
                                     out=torch.zeros((h, w, 4), dtype=torch.uint8))
    full_here = oracle.render(s, models, view, proj, w, h)["color"]
    assert np.array_equal(everywhere.numpy(), full_here) and np.array_equal(everywhere_w.numpy(), full_here)
    # frame-parallel bookkeeping: max-over-ranks timing as bench.py does it
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        full = oracle.render(s, models, view, proj, w, h)["color"]
        tall = [b - a for a, b in sh.all_bands(h, world, weights=wts)]
        ok = (frame is not None and np.array_equal(frame.numpy(), full) and again is out and np.array_equal(out.numpy(), full) and
              np.array_equal(weighted.numpy(), full) and tall[0] > max(tall[1:]))
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert frame is None and again is None and weighted is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sort_first_gather_gloo(tmp_path, orc, world):
    """Bands of equal (two ranks) and of different heights (three ranks), sent at their own sizes."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
