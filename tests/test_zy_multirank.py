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
           "--config", str(config), "--scale", "0.25", "--prewarm-frames", "8", "--check", "--mode", "sortfirst"]
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
           "--config", "3", "--scale", "0.25", "--prewarm-frames", "8", "--watchdog-seconds", "25", "--mode", "sortfirst"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["steps"] == 6 and out["warmup"] == 2
    assert out["value"] > 0 and out["config"]["parallelism"] == "frame-parallel2"
    assert out["sortfirst"]["error"] and out["invalid"].startswith("rehearsal")


def _bench(world, extra, env_extra=None, timeout=600):
    env = dict(os.environ, SXE_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--scale", "0.25", "--prewarm-frames", "8"] + extra
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_split_of_config3_is_frame_parallel_with_sort_first_beside_it():
    """Without --mode, N ranks draw whole frames of config 3 each (the headline, weak scaling, with the
    kernel figures of whole frames), and the sort-first split of the same frames is a further key;
    the 8K frame of config 4 is split sort-first."""
    out = _bench(2, ["--config", "3", "--check"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "frame-parallel2"
    assert out["config"]["frames_per_step"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["check"] == {"color_equal": True, "depth_equal": True}
    assert out["roofline"]["kernel"] in ("k_raster_shade", "k_setup") and 0.0 < out["roofline"]["frac"] <= 1.0
    sf = out["sort_first"]
    assert sf["scaling"] == "strong" and sf["value"] > 0 and sf["steps"] == 6 and sf["parallelism"].startswith("sortfirst-bands2")
    assert sf["check"] == {"color_equal": True, "depth_equal": True} and len(sf["per_rank"]) == 2
    out4 = _bench(2, ["--config", "4"])
    assert out4["scaling"] == "strong" and out4["frame_parallel"]["scaling"] == "weak"


@pytest.mark.gpu
@pytest.mark.parametrize("failure", ["raise", "hang"])
def test_a_failing_exchange_beside_a_frame_parallel_headline(failure):
    """The sort-first measurement beside a frame-parallel headline fails: the headline is complete and
    goes out, the failure is recorded under the key."""
    out = _bench(2, ["--config", "3", "--watchdog-seconds", "25"], {"SXE_BENCH_TEST_FAIL": failure}, timeout=300)
    assert out["scaling"] == "weak" and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["sort_first"]["error"] and "value" not in out["sort_first"]
