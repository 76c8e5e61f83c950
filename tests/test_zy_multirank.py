"""The N-rank sort-first path of bench.py end to end on one GPU: several ranks share the card, gloo
carries the band exchange through host memory (SXE_BENCH_REHEARSAL), and the frame rank 0 assembles
from the gathered bands must be the oracle's.  What differs from a real multi-GPU run is the
transport (RCCL there) and that every rank has a GPU of its own."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world, config", [(2, 3), (3, 4)])
def test_sort_first_bench_with_ranks_sharing_one_gpu(world, config):
    env = dict(os.environ, SXE_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--config", str(config), "--scale", "0.25", "--prewarm-frames", "8", "--check"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = [l for l in run.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["scaling"] == "strong" and len(out["per_rank"]) == world
    assert out["check"] == {"color_equal": True, "depth_equal": True}
    assert sum(r["rows"] for r in out["per_rank"]) == int({3: 2160, 4: 4320}[config] * 0.25)  # the bands cover the (scaled) frame
    assert all(r["raster_shade_ms"] > 0 for r in out["per_rank"])
    # the band heights come out of the trial of equal bands before the warm-up (through host memory the
    # gather is slow: rank 0, which sends nothing, ends up with the tallest band)
    tune = out["band_tuning"]
    assert tune["fixed_ms"] > 0 and tune["gather_ms_per_frame_over_one_link"] > 0 and len(tune["weights"]) == world
    assert [b - a for a, b in out["bands"]] == [int(r["rows"]) for r in out["per_rank"]]
    assert tune["weights"][0] >= max(tune["weights"][1:])
    # the frame-parallel split is measured as a second key
    fp = out["frame_parallel"]
    assert fp["scaling"] == "weak" and fp["frames_per_step"] == world and fp["value"] > 0
    assert fp["steps"] == 6 and fp["warmup"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("failure", ["raise", "hang"])
def test_a_band_exchange_that_fails_leaves_the_frame_parallel_line(failure):
    """The exchange is the one part of an N-rank run that no single-GPU run exercises with its real
    transport: when it raises, or does not come back, the run still ends, with status 0 and the
    frame-parallel measurement (taken first, under the same contract) as its line."""
    world = 2
    env = dict(os.environ, SXE_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", SXE_BENCH_TEST_FAIL=failure)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--config", "3", "--scale", "0.25", "--prewarm-frames", "8", "--watchdog-seconds", "25"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["steps"] == 6 and out["warmup"] == 2
    assert out["value"] > 0 and out["config"]["parallelism"] == "frame-parallel2"
    assert out["sortfirst"]["error"]
