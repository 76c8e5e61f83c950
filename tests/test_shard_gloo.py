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
    # frame-parallel bookkeeping: max-over-ranks timing as bench.py does it
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        full = oracle.render(s, models, view, proj, w, h)["color"]
        ok = frame is not None and np.array_equal(frame.numpy(), full) and again is out and np.array_equal(out.numpy(), full)
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert frame is None and again is None
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
