"""Multi-rank runs of the independent units (poses, Reeds-Shepp pairs, planner queries): dealt to
the ranks as bench.py deals them, gathered, and compared bit for bit with one rank working the whole
list.  On CPU (gloo, world_size 2 and 3) the CPU oracle stands in for the kernels; the GPU variant
runs the kernels through the C ABI from two processes sharing the card."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from path_planner_amd import sharding as S
from path_planner_amd import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, *args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "sharded_worker.py"), *args]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert " OK" in r.stdout, r.stdout[-2000:]
    return r.stdout


def test_shards_tile_the_single_rank_streams():
    """Rank r's poses / pairs are the slice [r n, (r + 1) n) of the stream one rank draws, and the
    round-robin deal of queries is a partition."""
    n, world = 257, 3
    whole = W.random_poses(world * n, 512)
    parts = [S.pose_shard(n, 512, r) for r in range(world)]
    for k in range(3):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), whole[k])
    w0, w1 = W.rs_pairs(world * n)
    parts = [S.pair_shard(n, r) for r in range(world)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), w0)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), w1)
    ids = np.arange(100, 117)
    dealt = [S.query_shard(ids, r, world) for r in range(world)]
    assert sorted(np.concatenate(dealt).tolist()) == ids.tolist()
    assert max(map(len, dealt)) - min(map(len, dealt)) <= 1
    back = np.empty_like(ids)
    for r, d in enumerate(dealt):
        back[r::world] = d
    assert np.array_equal(back, ids)
    # dealt by goal: a partition into runs of equal length whose goal sets overlap in at most one
    # goal per neighbouring pair of ranks
    rng = np.random.default_rng(3)
    for n, ngoals, world in ((17, 4, 3), (100, 7, 4), (5, 9, 8), (64, 1, 2)):
        keys = np.stack([rng.integers(0, ngoals, n), np.zeros(n, np.int64)], axis=1)
        keys[:, 1] = keys[:, 0] * 3 + 1
        parts = [S.query_shard_by_goal(keys, r, world) for r in range(world)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        sets = [set(keys[p, 0].tolist()) for p in parts]
        distinct = len(set(keys[:, 0].tolist()))
        assert sum(map(len, sets)) <= distinct + world - 1
        for p in parts:
            assert np.array_equal(p, np.sort(p))
        back = S.gather_by_position(np.arange(n)[parts[0]] * 2, parts[0], 0, 1, n) if world == 1 else None
        assert back is None or len(back) == n


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_units_gloo_host_engine(world):
    out = _run(world, "--mode", "host")
    assert "poses=same" in out and "pairs=same" in out and "queries=same" in out and "total=same" in out
    assert "by_goal=same" in out


@pytest.mark.gpu
def test_sharded_units_device():
    out = _run(2, "--mode", "device", "--poses", "200000", "--pairs", "20000", "--queries", "33")
    assert "poses=same" in out and "pairs=same" in out and "queries=same" in out and "by_goal=same" in out


def _bench(args, env_extra, timeout=280):
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it: the parent starts two fresh ranks (on a
    one-GPU box they share the card and reduce over gloo, which the line says), relays the one JSON
    line and exits 0.  The line carries the strong and the weak form of the sharded queries with
    per-rank search counters, and the slab wavefront."""
    import json
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--poses", "2097152", "--queries", "256",
                "--map", "1024", "--goals", "8", "--cells", "2048"], {})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    ex = d["extra"]
    assert "watchdog" not in ex, ex
    q, qw = ex["queries"], ex["queries_weak"]
    assert q["scaling"] == "strong" and q["config"]["queries"] == 256 and q["config"]["deal"].startswith("whole goal")
    assert qw["scaling"] == "weak" and qw["config"]["queries"] == 512
    for leg in (q, qw):
        pr = leg["search"]["per_rank"]
        assert len(pr) == 2 and all(c["rounds"] > 0 and c["queries"] > 0 for c in pr)
        # eight goals dealt by group: a rank relaxes about half of the grids, not all of them
        assert all(c["goal_grids"] <= 5 for c in pr), pr
        assert sum(c["queries"] for c in pr) == leg["config"]["queries"]
    assert ex["wavefront"]["value"] > 0 and ex["wavefront"]["config"]["exchange_rounds"] >= 1
    import torch
    if torch.cuda.device_count() < 2:
        assert "rehearsal" in d["config"]


@pytest.mark.gpu
def test_bench_watchdog_exits_nonzero():
    """Extra legs that do not finish in time: the headline line still goes out, marked, and the run
    ends with a non-zero status instead of reporting success."""
    import json
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--poses", "1048576", "--queries", "256",
                "--map", "1024", "--goals", "8", "--cells", "2048", "--extra-timeout", "0.05"], {})
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] > 0 and "stopped after" in d["extra"]["watchdog"]


@pytest.mark.gpu
def test_two_rank_bench_line_carries_the_sharded_legs():
    """The default bench run with two ranks (gloo, both on the one card, sizes shrunk): one JSON line
    whose headline is the pose-sharded collision check and whose `extra` holds the query-sharded
    planner and the slab wavefront with its halo exchange."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--poses", "2097152", "--queries", "256", "--map", "1024", "--goals", "8", "--cells", "2048"]
    env = dict(os.environ, PP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["poses_per_gpu"] == 2097152
    ex = d["extra"]
    assert "watchdog" not in ex, ex
    assert ex["queries"]["value"] > 0 and ex["queries"]["search"]["other_errors"] == 0, ex["queries"]
    assert ex["queries"]["config"]["parallelism"] == "query-sharded x2"
    assert ex["wavefront"]["value"] > 0 and ex["wavefront"]["config"]["exchange_rounds"] >= 1, ex["wavefront"]
