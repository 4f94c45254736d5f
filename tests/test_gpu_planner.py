"""The planner entry point on the device vs the CPU oracle: closed-set expansion order and
grid indices bit-exact, path cost / coordinates to 1e-9 relative."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import api
from path_planner_amd import workloads as W

pytestmark = pytest.mark.gpu
CONF = A.demo_config()


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def same_plan(r, e):
    """True if the two runs agree; a disagreement is tolerated (False) only when the device
    run reported a decision inside its guard band."""
    try:
        _same_plan(r, e)
        return True
    except AssertionError:
        if r["stats"].ambiguous:
            return False
        raise


def _same_plan(r, e):
    assert r["status"] == e["status"]
    S, E = r["stats"], e["stats"]
    assert (r["closed"] == e["closed"]).all()          # expansion order, bit-exact
    assert S.expansions == E.expansions and S.stale_pops == E.stale_pops
    assert S.rollouts == E.rollouts and S.rs_paths == E.rs_paths
    assert S.poses_checked == E.poses_checked
    if r["status"] == 0:
        assert len(r["x"]) == len(e["x"])
        assert abs(r["cost"] - e["cost"]) <= 1e-9 * abs(e["cost"])
        assert np.allclose(r["x"], e["x"], rtol=1e-9, atol=1e-9)
        assert np.allclose(r["y"], e["y"], rtol=1e-9, atol=1e-9)
        assert np.allclose(r["yaw"], e["yaw"], rtol=1e-9, atol=1e-9)
        assert (r["dir"] == e["dir"]).all()
        assert abs(S.rs_final_length - E.rs_final_length) <= 1e-9 * E.rs_final_length


def test_demo_query(ctx):
    """src/main.rs:215-269."""
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    (sx, sy, syaw), (gx, gy, gyaw) = W.demo_query()
    r = api.hybrid_astar_planning(ctx, m, CONF, sx, sy, syaw, gx, gy, gyaw)
    e = O.OracleMap(ox, oy).plan(CONF, [sx, sy, syaw], [gx, gy, gyaw])
    assert same_plan(r, e)
    assert r["stats"].expansions == 24 and r["stats"].stale_pops == 11
    assert r["closed"][:12].tolist() == [7980, 8004, 7304, 6604, 5881, 4758, 4760, 4361, 4736, 3986, 3563, 3539]
    assert abs(r["cost"] - 561.4479147109321) < 1e-9 * 561.45
    assert len(r["x"]) == 184 and r["stats"].poses_checked == 8951


def test_random_queries_on_demo_map(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    om = O.OracleMap(ox, oy)
    rng = np.random.default_rng(12)
    checked = nopath = 0
    # unreachable goals never terminate in the reference (unwrapped yaw): cap both sides
    ctx.set_max_expansions(1500)
    O.set_max_expansions(1500)
    for _ in range(25):
        s = [rng.uniform(5, 45), rng.uniform(5, 25), rng.uniform(-math.pi, math.pi)]
        g = [rng.uniform(5, 45), rng.uniform(5, 25), rng.uniform(-math.pi, math.pi)]
        r = api.hybrid_astar_planning(ctx, m, CONF, *s, *g)
        e = om.plan(CONF, s, g)
        checked += bool(same_plan(r, e))
        nopath += r["status"] != 0
    ctx.set_max_expansions(100000)
    O.set_max_expansions(0)
    assert checked >= 22 and nopath < 20


def test_plan_batch_matches_single_and_oracle(ctx):
    occ = W.occupancy_grid(96, 0.004, seed=99)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    rng = np.random.default_rng(3)
    n = 40
    goals_pool = np.c_[rng.uniform(15, 80, 4), rng.uniform(15, 80, 4), rng.uniform(-3, 3, 4)]
    starts = np.c_[rng.uniform(12, 84, n), rng.uniform(12, 84, n), rng.uniform(-3, 3, n)]
    goals = goals_pool[rng.integers(0, 4, n)]
    lib = ctx.lib
    ctx.set_max_expansions(1500)
    O.set_max_expansions(1500)
    cap, ccap = 20000, 1 << 16
    X = np.zeros((n, cap)); Y = np.zeros((n, cap)); YAW = np.zeros((n, cap))
    D = np.zeros((n, cap), np.int64)
    closed = np.zeros((n, ccap), np.uint64)
    paths = (A.Path * n)()
    stats = (A.PlanStats * n)()
    for i in range(n):
        paths[i] = A.Path(X[i].ctypes.data_as(A.c_double_p), Y[i].ctypes.data_as(A.c_double_p),
                          YAW[i].ctypes.data_as(A.c_double_p), D[i].ctypes.data_as(A.c_i64_p), cap, 0, 0.0)
        stats[i].closed_cap = ccap
        stats[i].closed_order = closed[i].ctypes.data_as(A.c_u64_p)
    status = np.zeros(n, np.int32)
    st = lib.pp_plan_batch(ctx.h, m.h, C.byref(CONF), n, np.ascontiguousarray(starts).ctypes.data_as(A.c_double_p),
                           np.ascontiguousarray(goals).ctypes.data_as(A.c_double_p), paths, stats,
                           status.ctypes.data_as(A.c_i32_p))
    assert st == 0
    ok = 0
    for i in range(n):
        e = om.plan(CONF, starts[i], goals[i])
        assert status[i] == e["status"]
        ne = int(stats[i].expansions)
        if stats[i].ambiguous and (ne != e["stats"].expansions or (closed[i, :ne] != e["closed"]).any()):
            continue
        assert ne == e["stats"].expansions
        assert (closed[i, :ne] == e["closed"]).all()
        if status[i] == 0:
            L = int(paths[i].len)
            assert L == len(e["x"]) and abs(paths[i].cost - e["cost"]) <= 1e-9 * e["cost"]
            assert np.allclose(X[i, :L], e["x"], rtol=1e-9, atol=1e-9)
            ok += 1
    ctx.set_max_expansions(100000)
    O.set_max_expansions(0)
    assert ok >= n // 2


def test_plan_error_paths(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    # goal outside the heuristic grid: the reference panics on hmap indexing
    with pytest.raises(api.PlannerError) as e:
        api.hybrid_astar_planning(ctx, m, CONF, 10.0, 7.0, 0.0, 60.0, 20.0, 0.0)
    assert e.value.status == A.PP_ERR_BOUNDS
    # start == goal: Reeds-Shepp assert
    with pytest.raises(api.PlannerError) as e:
        api.hybrid_astar_planning(ctx, m, CONF, 10.0, 7.0, 0.5, 10.0, 7.0, 0.5)
    assert e.value.status == A.PP_ERR_DEGENERATE_RS
    # unreachable goal: the reference searches forever (unwrapped yaw); the cap reports it
    ctx.set_max_expansions(300)
    O.set_max_expansions(300)
    r = api.hybrid_astar_planning(ctx, m, CONF, 25.0, 8.0, 0.0, 45.0, 28.5, 0.0)
    eo = O.OracleMap(ox, oy).plan(CONF, [25.0, 8.0, 0.0], [45.0, 28.5, 0.0])
    ctx.set_max_expansions(100000)
    O.set_max_expansions(0)
    assert r["status"] == eo["status"] == A.PP_ERR_LIMIT
    assert (r["closed"][:300] == eo["closed"][:300]).all() or r["stats"].ambiguous


def test_plan_batch_many_queries_on_a_large_map_vs_oracle(ctx):
    """The many-queries workload at reduced size: 96 queries sharing 8 goals on a 1024 x 1024 map
    with clustered wall segments; every query against the CPU oracle."""
    occ = W.plan_map(1024)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    cs, cg = W.plan_queries(400, 1024, 8)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:96]
    assert len(keep) == 96
    starts, goals = cs[keep], cg[keep]
    ctx.set_max_expansions(2000)
    O.set_max_expansions(2000)
    try:
        r = api.plan_batch(ctx, m, CONF, starts, goals, cap=4096, closed_cap=2048)
        bad = 0
        for i in range(96):
            e = om.plan(CONF, starts[i], goals[i])
            ne = int(r["stats"][i].expansions)
            same = (r["status"][i] == e["status"] and ne == e["stats"].expansions and
                    (r["closed"][i, :min(ne, 2048)] == e["closed"][:2048]).all())
            if same and e["status"] == 0:
                L = int(r["len"][i])
                same = (L == len(e["x"]) and abs(r["cost"][i] - e["cost"]) <= 1e-9 * abs(e["cost"]) and
                        np.allclose(r["x"][i, :L], e["x"], rtol=1e-9, atol=1e-9) and
                        np.allclose(r["yaw"][i, :L], e["yaw"], rtol=1e-9, atol=1e-9) and
                        (r["dir"][i, :L] == e["dir"]).all())
            if not same:
                assert r["stats"][i].ambiguous, i  # only a guard-band decision may differ
                bad += 1
        assert bad <= 5
        assert (r["status"] == 0).sum() >= 80
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)


def test_bounded_heuristic_grids_do_not_change_a_search(ctx):
    """The planner relaxes its heuristic grids only up to a cost limit and completes them when a
    search looks beyond it.  Whatever the limit (PP_HEURISTIC_LIMIT overrides the planner's own
    choice), closing order, counters and paths are those of the run on complete grids."""
    occ = W.plan_map(1024)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(300, 1024, 6)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:64]
    starts, goals = cs[keep], cg[keep]
    ctx.set_max_expansions(1500)
    runs = {}
    try:
        for lim in ("inf", None, "40", "3", "0"):
            if lim is None:
                os.environ.pop("PP_HEURISTIC_LIMIT", None)
            else:
                os.environ["PP_HEURISTIC_LIMIT"] = lim
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=4096, closed_cap=1500)
            runs[lim] = {k: np.array(r[k], copy=True) for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")}
            runs[lim]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
            runs[lim]["poses"] = np.array([int(s.poses_checked) for s in r["stats"]])
    finally:
        os.environ.pop("PP_HEURISTIC_LIMIT", None)
        ctx.set_max_expansions(100000)
    ref = runs["inf"]
    assert (ref["status"] == 0).sum() >= 50 and ref["exp"].max() > 20
    for lim, r in runs.items():
        for k, v in ref.items():
            assert np.array_equal(v, r[k], equal_nan=True), (lim, k)
