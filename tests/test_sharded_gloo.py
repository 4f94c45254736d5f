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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_units_gloo_host_engine(world):
    out = _run(world, "--mode", "host")
    assert "poses=same" in out and "pairs=same" in out and "queries=same" in out and "total=same" in out


@pytest.mark.gpu
def test_sharded_units_device():
    out = _run(2, "--mode", "device", "--poses", "200000", "--pairs", "20000", "--queries", "33")
    assert "poses=same" in out and "pairs=same" in out and "queries=same" in out


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
