"""Parity of the plain grid A* entry points against the CPU oracle: closing order and path
bit-exact (coordinates are cell indices times reso)."""
import numpy as np
import pytest

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import api
from path_planner_amd import workloads as W
from test_oracle_astar import astar_env, check_astar_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def same(ctx, m, ox, oy, s, g, reso, rr):
    st, exp, exp_order = O.astar(ox, oy, s[0], s[1], g[0], g[1], reso, rr)
    got, order = api.astar_planning(ctx, m, s[0], s[1], g[0], g[1], reso, rr, closed_cap=1000000)
    assert order.tolist() == exp_order.tolist()
    if exp is None:
        assert st == A.PP_NO_PATH and got is None
    else:
        assert (got[0] == exp[0]).all() and (got[1] == exp[1]).all()
    return exp


def test_against_the_patched_twin_fixture(ctx):
    """tests/golden/astar_rust.json (the reference's Python twin with the crate's rounding and
    heap tie rule)."""
    def plan(ox, oy, sx, sy, gx, gy, reso, rr):
        m = api.ObstacleMap(ctx, ox, oy)
        got, order = api.astar_planning(ctx, m, sx, sy, gx, gy, reso, rr, closed_cap=1000000)
        return (A.PP_NO_PATH if got is None else 0), got, order
    check_astar_fixture(plan)


def test_demo_environment(ctx):
    ox, oy = astar_env()
    m = api.ObstacleMap(ctx, ox, oy)
    path = same(ctx, m, ox, oy, (10.0, 10.0), (50.0, 50.0), 1.0, 2.0)  # main.rs:157-171
    assert path is not None and len(path[0]) > 60


def test_edge_cases(ctx):
    ox, oy = astar_env()
    m = api.ObstacleMap(ctx, ox, oy)
    assert len(same(ctx, m, ox, oy, (30.0, 30.0), (30.0, 30.0), 1.0, 2.0)[0]) == 2   # start == goal
    assert same(ctx, m, ox, oy, (10.0, 10.0), (20.0, 20.0), 1.0, 2.0) is None        # goal blocked
    assert same(ctx, m, ox, oy, (100.0, 100.0), (50.0, 50.0), 1.0, 2.0) is None      # start outside
    assert same(ctx, m, ox, oy, (-5.0, 30.0), (50.0, 50.0), 1.0, 2.0) is None        # start clamps to 0
    assert same(ctx, m, ox, oy, (10.0, 10.0), (50.0, 50.0), 1.0, 0.3) is not None    # thin walls
    same(ctx, m, ox, oy, (10.4, 10.6), (50.5, 49.5), 2.0, 2.0)                       # .5 ties round away
    # capacity: the required length is reported
    n = __import__("ctypes").c_uint64(0)
    st = ctx.lib.pp_astar_plan(ctx.h, m.h, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0, None, None, 0,
                               __import__("ctypes").byref(n), None, 0, None)
    assert st == A.PP_ERR_CAPACITY and n.value > 60


def test_random_maps(ctx):
    rng = np.random.default_rng(11)
    for trial in range(10):
        n = int(rng.integers(30, 260))
        reso = [1.0, 2.0, 0.5][trial % 3]
        ox, oy = W.raster_to_points(W.occupancy_grid(n, 0.2, seed=500 + trial))
        ox, oy = ox * reso, oy * reso
        m = api.ObstacleMap(ctx, ox, oy)
        for _ in range(4):
            s = rng.uniform(1, n - 1, 2) * reso
            g = rng.uniform(1, n - 1, 2) * reso
            same(ctx, m, ox, oy, s, g, reso, 0.45 * reso)


def test_scattered_obstacle_points(ctx):
    """Obstacles off the lattice (the raster comes from the general point path)."""
    rng = np.random.default_rng(5)
    ox = np.concatenate([rng.uniform(0, 80, 500), [0.0, 80.0]])
    oy = np.concatenate([rng.uniform(0, 80, 500), [0.0, 80.0]])
    m = api.ObstacleMap(ctx, ox, oy)
    found = 0
    for _ in range(8):
        s, g = rng.uniform(5, 75, 2), rng.uniform(5, 75, 2)
        found += same(ctx, m, ox, oy, s, g, 1.0, 1.2) is not None
    assert found >= 2


def test_batch_equals_singles(ctx):
    n = 400
    ox, oy = W.raster_to_points(W.occupancy_grid(n, 0.15, seed=77))
    m = api.ObstacleMap(ctx, ox, oy)
    rng = np.random.default_rng(3)
    starts, goals = rng.uniform(1, n - 1, (24, 2)), rng.uniform(1, n - 1, (24, 2))
    paths, ex = api.astar_planning_batch(ctx, m, starts, goals, 1.0, 0.45)
    some = 0
    for i in range(24):
        one = api.astar_planning(ctx, m, *starts[i], *goals[i], 1.0, 0.45)
        assert (one is None) == (paths[i] is None)
        if one is not None:
            some += 1
            assert (one[0] == paths[i][0]).all() and (one[1] == paths[i][1]).all()
    assert some >= 5 and (ex > 0).all()
    st, exp, order = O.astar(ox, oy, *starts[0], *goals[0], 1.0, 0.45)
    assert ex[0] == len(order)
