"""Multi-rank tiled heuristic: the exchange protocol on CPU (gloo, world_size 2 and 3) with a host
stand-in for the slab kernel, and the slab kernel itself on the GPU under the same driver.  Each
case is compared bit for bit with the CPU oracle's Dijkstra by rank 0 of the worker."""
import os
import socket
import subprocess
import sys

import pytest

from path_planner_amd import tiled

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, *args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "tiled_worker.py"), *args]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert " OK" in r.stdout, r.stdout[-2000:]
    return r.stdout


def test_slab_bounds_cover_the_grid():
    for xw in (1, 7, 64, 16384):
        for world in (1, 2, 3, 8):
            if world > xw:
                continue
            b = [tiled.slab_bounds(xw, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == xw
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


@pytest.mark.parametrize("world,extra", [
    (2, []),
    (3, ["--xw", "50", "--yw", "33", "--density", "0.3", "--seed", "9"]),
    (2, ["--density", "0.0", "--xw", "48", "--yw", "30"]),          # walls: many exchange rounds
    (2, ["--goal", "1", "1", "--xw", "40", "--yw", "40", "--density", "0.15"]),
    (3, ["--band", "1.0", "--xw", "45", "--yw", "31"]),            # a band below one diagonal step
    (2, ["--band", "1e9", "--density", "0.0", "--xw", "48", "--yw", "30"]),  # every slab to its fixpoint
])
def test_tiled_protocol_gloo_host_solver(world, extra):
    out = _run(world, "--mode", "host", *extra)
    assert "rounds=" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,extra", [
    (2, ["--xw", "300", "--yw", "257", "--density", "0.25"]),
    (3, ["--xw", "200", "--yw", "190", "--density", "0.0"]),
    (2, ["--xw", "700", "--yw", "300", "--density", "0.2", "--band", "64"]),
])
def test_tiled_slab_kernel_device(world, extra):
    _run(world, "--mode", "device", *extra)
