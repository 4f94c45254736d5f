"""The planner entry point on the device vs the CPU oracle: closed-set expansion order and
grid indices bit-exact, path cost / coordinates to 1e-9 relative."""
import ctypes as C
import math

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


def explained_by_guard_band(stats, closed, exp_closed):
    """The bound a differing query is held to: the device reported a guard-band decision, and
    everything it closed up to and including the node whose expansion took the first such decision
    is the oracle's closing order (a different last bit in a libm can only act from there on)."""
    assert stats.ambiguous > 0, "differs from the oracle without any guard-band decision"
    k = int(stats.first_ambiguous) + 1
    assert 1 <= k <= int(stats.expansions)
    k = min(k, len(closed), len(exp_closed))
    assert (np.asarray(closed[:k]) == np.asarray(exp_closed[:k])).all(), "differs before the first guard-band decision"


def same_plan(r, e):
    """True if the two runs agree in everything; False if they differ in a way the device's first
    guard-band decision accounts for; raises otherwise."""
    try:
        _same_plan(r, e)
        return True
    except AssertionError:
        explained_by_guard_band(r["stats"], r["closed"], e["closed"])
        return False


def test_no_guard_band_means_unset_position(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    r = api.hybrid_astar_planning(ctx, m, CONF, 30.0, 6.5, 0.3, 35.0, 7.0, 0.2)
    if r["stats"].ambiguous == 0:
        assert r["stats"].first_ambiguous == 2**64 - 1
    else:
        assert r["stats"].first_ambiguous < r["stats"].expansions


def test_demo_and_random_queries_against_the_patched_twin(ctx):
    """tests/golden/rust_semantics.json: the demo query of src/main.rs:215-269 and 24 random
    demo-map queries from the reference's Python twin patched to the crate's semantics."""
    from test_oracle_golden import check_rust_semantics_fixture
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    data, flagged = check_rust_semantics_fixture(
        lambda s, g: api.hybrid_astar_planning(ctx, m, CONF, *s, *g), strict=False)
    for qi, q, r in flagged:
        explained_by_guard_band(r["stats"], r["closed"], q["closed"])
    # the demo query itself takes no decision that a libm's last bit could turn
    assert 0 not in [qi for qi, _, _ in flagged]
    assert len(flagged) <= FLAGGED_FIXTURE, [qi for qi, _, _ in flagged]


# Queries (fixed seeds) whose device run differs from the oracle's after a guard-band decision;
# each of them passed explained_by_guard_band.  The numbers are what this build measures on an
# MI355X; a change in them means a kernel's arithmetic changed and wants a look.
FLAGGED_FIXTURE = 0
FLAGGED_DEMO_RANDOM = 0
FLAGGED_BATCH40 = 0
FLAGGED_BATCH96 = 0
FLAGGED_TWO_LANES = 0    # the 72 sampled queries of the 1500-query batch
FLAGGED_SEVERAL_PASSES = 0


def test_random_queries_on_demo_map(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    om = O.OracleMap(ox, oy)
    rng = np.random.default_rng(12)
    differ = nopath = 0
    # unreachable goals never terminate in the reference (unwrapped yaw): cap both sides
    ctx.set_max_expansions(1500)
    O.set_max_expansions(1500)
    try:
        for _ in range(25):
            s = [rng.uniform(5, 45), rng.uniform(5, 25), rng.uniform(-math.pi, math.pi)]
            g = [rng.uniform(5, 45), rng.uniform(5, 25), rng.uniform(-math.pi, math.pi)]
            r = api.hybrid_astar_planning(ctx, m, CONF, *s, *g)
            e = om.plan(CONF, s, g)
            differ += not same_plan(r, e)
            nopath += r["status"] != 0
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    print("differing after a guard-band decision:", differ)
    assert differ <= FLAGGED_DEMO_RANDOM, differ
    assert nopath < 20


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
    ctx.set_max_expansions(1500)
    O.set_max_expansions(1500)
    try:
        r = api.plan_batch(ctx, m, CONF, starts, goals, cap=20000, closed_cap=1 << 12)
        differ = found = 0
        for i in range(n):
            e = om.plan(CONF, starts[i], goals[i])
            differ += not same_batch_query(r, i, e, 1 << 12)
            found += r["status"][i] == 0
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    print("differing after a guard-band decision:", differ)
    assert differ <= FLAGGED_BATCH40, differ
    assert found >= n // 2


def same_batch_query(r, i, e, closed_cap):
    """Query i of a plan_batch result against the oracle's run: same_plan's three outcomes."""
    S = r["stats"][i]
    ne = int(S.expansions)
    closed = r["closed"][i, :min(ne, closed_cap)]
    E = e["stats"]
    same = (r["status"][i] == e["status"] and ne == E.expansions and
            (closed == e["closed"][:closed_cap]).all() and S.stale_pops == E.stale_pops and
            S.rollouts == E.rollouts and S.rs_paths == E.rs_paths and S.poses_checked == E.poses_checked)
    if same and e["status"] == 0:
        L = int(r["len"][i])
        same = (L == len(e["x"]) and abs(r["cost"][i] - e["cost"]) <= 1e-9 * abs(e["cost"]) and
                np.allclose(r["x"][i, :L], e["x"], rtol=1e-9, atol=1e-9) and
                np.allclose(r["y"][i, :L], e["y"], rtol=1e-9, atol=1e-9) and
                np.allclose(r["yaw"][i, :L], e["yaw"], rtol=1e-9, atol=1e-9) and
                (r["dir"][i, :L] == e["dir"]).all())
    if not same:
        explained_by_guard_band(S, closed, e["closed"])
    return same


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
    if not (r["closed"][:300] == eo["closed"][:300]).all():
        explained_by_guard_band(r["stats"], r["closed"], eo["closed"])


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
        differ = 0
        for i in range(96):
            e = om.plan(CONF, starts[i], goals[i])
            differ += not same_batch_query(r, i, e, 2048)
        print("differing after a guard-band decision:", differ)
        assert differ <= FLAGGED_BATCH96, differ
        assert (r["status"] == 0).sum() >= 80
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)


def test_large_batch_in_two_lanes_equals_small_batches_and_oracle(ctx):
    """From 1024 queries on a batch runs as two lanes, on a stream each, that alternate on the host.
    Every query must come out as it does in batches below that size (one lane), and a sample of them
    as the CPU oracle has it; the map is dense enough for many long searches."""
    occ = W.plan_map(1024, 0.02)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(6000, 1024, 16, min_dist=20.0, max_dist=150.0)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:1500]
    assert len(keep) == 1500
    starts, goals = cs[keep], cg[keep]
    ctx.set_max_expansions(400)
    O.set_max_expansions(400)
    try:
        big = api.plan_batch(ctx, m, CONF, starts, goals, cap=2048, closed_cap=400)
        big = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in big.items()}
        big_exp = np.array([int(s.expansions) for s in big["stats"]])
        assert (big_exp > 40).sum() > 100, "too few long searches"
        for lo in range(0, 1500, 500):
            part = api.plan_batch(ctx, m, CONF, starts[lo:lo + 500], goals[lo:lo + 500], cap=2048, closed_cap=400)
            sl = slice(lo, lo + 500)
            for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed"):
                assert np.array_equal(big[k][sl], part[k], equal_nan=True), (lo, k)
            assert np.array_equal(big_exp[sl], [int(s.expansions) for s in part["stats"]])
        om = O.OracleMap(ox, oy)
        differ = 0
        # (searches still running after 64 rounds have their rollouts pre-filtered on the device by
        # tables of the cells they have closed or hold open: the longest ones exercise that)
        for i in list(range(0, 1500, 47)) + list(np.argsort(big_exp)[-40:]):
            e = om.plan(CONF, starts[i], goals[i])
            differ += not same_batch_query(big, int(i), e, 400)
        print("differing after a guard-band decision:", differ)
        assert differ <= FLAGGED_TWO_LANES, differ
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)


def test_two_lanes_with_failing_queries(ctx):
    """A large batch on the demo map in which most queries fail one way or another (goal outside
    the heuristic grid, start on the goal, start in collision, expansion cap): per-query status,
    counters and closing order as in batches small enough for one lane."""
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    rng = np.random.default_rng(31)
    n = 1100
    starts = np.column_stack([rng.uniform(4, 46, n), rng.uniform(4, 26, n), rng.uniform(-3, 3, n)])
    goals = np.column_stack([rng.uniform(4, 46, n), rng.uniform(4, 26, n), rng.uniform(-3, 3, n)])
    goals[::7] = [60.0, 20.0, 0.0]            # outside the grid: the reference panics on hmap indexing
    goals[3::11] = starts[3::11]              # start == goal: Reeds-Shepp assert
    starts[5::13] = [25.0, 8.0, 0.0]          # unreachable goal behind the wall: runs into the cap
    goals[5::13] = [45.0, 28.5, 0.0]
    ctx.set_max_expansions(120)
    try:
        big = api.plan_batch(ctx, m, CONF, starts, goals, cap=1024, closed_cap=120)
        exp_big = np.array([int(s.expansions) for s in big["stats"]])
        seen = set(int(v) for v in big["status"])
        assert {A.PP_OK, A.PP_ERR_BOUNDS, A.PP_ERR_DEGENERATE_RS, A.PP_ERR_LIMIT} <= seen, seen
        for lo in range(0, n, 275):
            part = api.plan_batch(ctx, m, CONF, starts[lo:lo + 275], goals[lo:lo + 275], cap=1024, closed_cap=120)
            sl = slice(lo, lo + 275)
            for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed"):
                assert np.array_equal(big[k][sl], part[k], equal_nan=True), (lo, k)
            assert np.array_equal(exp_big[sl], [int(s.expansions) for s in part["stats"]])
    finally:
        ctx.set_max_expansions(100000)


def test_bounded_heuristic_grids_do_not_change_a_search(ctx):
    """The planner relaxes its heuristic grids only up to a cost limit and completes them when a
    search looks beyond it.  Whatever the limit (pp_ctx_set_heuristic_limit overrides the planner's
    own choice), closing order, counters and paths are those of the run on complete grids."""
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
        for lim in (float("inf"), None, 40.0, 3.0, 0.0):
            ctx.set_heuristic_limit(-1.0 if lim is None else lim)
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=4096, closed_cap=1500)
            runs[lim] = {k: np.array(r[k], copy=True) for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")}
            runs[lim]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
            runs[lim]["poses"] = np.array([int(s.poses_checked) for s in r["stats"]])
    finally:
        ctx.set_heuristic_limit(-1.0)
        ctx.set_max_expansions(100000)
    ref = runs[float("inf")]
    assert (ref["status"] == 0).sum() >= 50 and ref["exp"].max() > 20
    for lim, r in runs.items():
        for k, v in ref.items():
            assert np.array_equal(v, r[k], equal_nan=True), (lim, k)


def test_heuristic_grids_in_several_passes(ctx):
    """With a small memory budget (pp_ctx_set_heuristic_workspace) the heuristic grids of a batch's
    goal cells are relaxed a few at a time and kept as dense arrays instead of the relaxation's own
    layout: closing order, counters and paths are those of the one-pass run, and of the oracle."""
    occ = W.plan_map(512, 0.01)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(400, 512, 7, min_dist=10.0, max_dist=120.0)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 503) & (cs[:, 1] > 8) & (cs[:, 1] < 503)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:96]
    starts, goals = cs[keep], cg[keep]
    ngoals = len({(round(g[0] / CONF.xy_reso), round(g[1] / CONF.xy_reso)) for g in goals})
    assert ngoals >= 5
    ctx.set_max_expansions(800)
    O.set_max_expansions(800)
    runs = {}
    try:
        # a 256 x 256-cell grid with its separator column is budgeted at 10 x 8 blocks of 8 KiB:
        # room for all grids / for three at a time / for one
        for budget in (8 << 30, 3 * 80 * 8192, 8192):
            ctx.set_heuristic_workspace(budget)
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=2048, closed_cap=800)
            runs[budget] = {k: np.array(r[k], copy=True) for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")}
            runs[budget]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
            runs[budget]["poses"] = np.array([int(s.poses_checked) for s in r["stats"]])
            if budget == 8192:
                last = r
        ref = runs[8 << 30]
        assert (ref["status"] == 0).sum() >= 60 and ref["exp"].max() > 10
        for budget, r in runs.items():
            for k, v in ref.items():
                assert np.array_equal(v, r[k], equal_nan=True), (budget, k)
        om = O.OracleMap(ox, oy)
        differ = 0
        for i in range(0, len(starts), 8):
            e = om.plan(CONF, starts[i], goals[i])
            differ += not same_batch_query(last, int(i), e, 800)
        print("differing after a guard-band decision:", differ)
        assert differ <= FLAGGED_SEVERAL_PASSES, differ
    finally:
        ctx.set_heuristic_workspace(8 << 30)
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)


def test_table_budget_does_not_change_a_batch(ctx):
    """The device-side tables of known cells (searches still running after 64 rounds) only spare the
    host work.  With the default budget, with one that forces small tables and with none at all
    (pp_ctx_set_table_budget; the same path a device without room for them takes) every query comes
    out the same."""
    occ = W.plan_map(1024, 0.02)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(6000, 1024, 16, min_dist=20.0, max_dist=150.0)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:1400]
    starts, goals = cs[keep], cg[keep]
    ctx.set_max_expansions(300)
    runs = []
    try:
        # 2^17 slots of 16 bytes per live query by default; 48 MiB leaves the hundred-odd queries still
        # alive after 64 rounds 16384 slots each; 0 switches the tables off
        for budget in (16 << 30, 48 << 20, 0):
            ctx.set_table_budget(budget)
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=2048, closed_cap=300)
            runs.append({k: np.array(r[k], copy=True) for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")})
            runs[-1]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
            runs[-1]["poses"] = np.array([int(s.poses_checked) for s in r["stats"]])
    finally:
        ctx.set_table_budget(16 << 30)
        ctx.set_max_expansions(100000)
    # enough searches outlive round 64 for the tables to be handed out
    assert (runs[0]["exp"] > 80).sum() >= 64
    for r in runs[1:]:
        for k, v in runs[0].items():
            assert np.array_equal(v, r[k], equal_nan=True), k


def test_two_lanes_with_one_emptied_early_equal_one_lane(ctx):
    """The two lanes of a large batch run on streams of their own and share the tables of known cells
    and the heuristic grids.  A batch dealt so that one lane (the even positions) holds only short
    searches -- it is done long before the other, which then goes on alone, several nodes per query
    and round, tables in use -- comes out as its halves do when planned as batches of one lane."""
    occ = W.plan_map(1024, 0.02)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(6000, 1024, 16, min_dist=20.0, max_dist=150.0)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:1200]
    starts, goals = cs[keep], cg[keep]
    keys = ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")

    def run(st, go):
        r = api.plan_batch(ctx, m, CONF, st, go, cap=2048, closed_cap=1500)
        out = {k: np.array(r[k], copy=True) for k in keys}
        out["exp"] = np.array([int(s.expansions) for s in r["stats"]])
        out["poses"] = np.array([int(s.poses_checked) for s in r["stats"]])
        out["lanes"] = api.plan_batch_counters(ctx)["lanes"]
        return out

    ctx.set_max_expansions(1500)
    try:
        halves = [run(starts[lo:lo + 600], goals[lo:lo + 600]) for lo in (0, 600)]
        assert all(h["lanes"] == 1 for h in halves)
        ref = {k: np.concatenate([h[k] for h in halves]) for k in keys + ("exp", "poses")}
        assert (ref["exp"] > 300).sum() >= 8, "too few long searches"
        both = run(starts, goals)
        assert both["lanes"] == 2
        for k, v in ref.items():
            assert np.array_equal(v, both[k], equal_nan=True), k
        order = np.argsort(ref["exp"], kind="stable")
        half = len(order) // 2
        perm = np.empty(2 * half, dtype=np.int64)
        perm[0::2], perm[1::2] = order[:half], order[len(order) - half:]
        assert int(ref["exp"][order[half - 1]]) < 60
        dealt = run(starts[perm], goals[perm])
        for k, v in ref.items():
            assert np.array_equal(v[perm], dealt[k], equal_nan=True), k
        # heuristic grids relaxed to a low cost limit only: a lane whose search looks beyond it has
        # the grids completed while the other lane's round is in flight on its own stream
        for lim in (3.0, 40.0):
            ctx.set_heuristic_limit(lim)
            bounded = run(starts, goals)
            assert bounded["lanes"] == 2
            for k, v in ref.items():
                assert np.array_equal(v, bounded[k], equal_nan=True), (lim, k)
    finally:
        ctx.set_heuristic_limit(-1.0)
        ctx.set_max_expansions(100000)


def test_batch_over_several_contexts_equals_one_context(ctx):
    """pp_plan_batch_multi deals the queries over contexts (one per device in a multi-GPU process;
    here three on the one card) by goal group, a host thread each, and gathers every result at its
    query's place: equal to pp_plan_batch over the whole list, failing and capped queries included."""
    occ = W.plan_map(1024, 0.02)
    cs, cg = W.plan_queries(3000, 1024, 9, min_dist=20.0, max_dist=150.0)
    m = api.ObstacleMap(ctx, raster=occ)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:397]
    starts, goals = cs[keep].copy(), cg[keep].copy()
    starts[5] = [-40.0, 20.0, 0.0]      # outside the map: no successor passes the index test, None
    starts[11] = goals[11]              # start == goal: the Reeds-Shepp assert
    ctxs = [api.Context(0) for _ in range(3)]
    maps = [api.ObstacleMap(c, raster=occ) for c in ctxs]
    keys = ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")
    try:
        for c in ctxs + [ctx]:
            c.set_max_expansions(200)
            c.set_host_threads(4)
        one = api.plan_batch(ctx, m, CONF, starts, goals, cap=2048, closed_cap=200)
        one = {k: np.array(one[k], copy=True) for k in keys} | {"exp": [int(s.expansions) for s in one["stats"]]}
        many = api.plan_batch(ctxs, maps, CONF, starts, goals, cap=2048, closed_cap=200)
        for k in keys:
            assert np.array_equal(one[k], many[k], equal_nan=True), k
        assert one["exp"] == [int(s.expansions) for s in many["stats"]]
        seen = set(int(v) for v in one["status"])
        assert {A.PP_OK, A.PP_NO_PATH, A.PP_ERR_LIMIT, A.PP_ERR_DEGENERATE_RS} <= seen, seen
        # dealt by goal: a context relaxed about a third of the nine goals' grids, not all of them
        grids = [api.plan_batch_counters(c)["goal_grids"] for c in ctxs]
        assert sum(api.plan_batch_counters(c)["queries"] for c in ctxs) == len(starts)
        assert max(grids) <= 5 and sum(grids) <= 9 + 2 + 2, grids
        # fewer queries than contexts, and the same context twice
        few = api.plan_batch(ctxs, maps, CONF, starts[:2], goals[:2], cap=2048, closed_cap=200)
        assert np.array_equal(few["status"], one["status"][:2]) and np.array_equal(few["cost"], one["cost"][:2], equal_nan=True)
        with pytest.raises(api.PlannerError) as e:
            api.plan_batch([ctxs[0], ctxs[0]], maps[:2], CONF, starts[:4], goals[:4])
        assert e.value.status == A.PP_ERR_INVALID
    finally:
        ctx.set_max_expansions(100000)
        ctx.set_host_threads(15)
        for o in maps + ctxs:
            o.close()


def test_batch_planned_in_parts_equals_unsplit(ctx):
    """From 2048 queries on pp_plan_batch plans a batch in up to four parts on streams of their own
    (pp_ctx_set_batch_parts: 0 = that choice, 1 = never, n = always n).  Every query comes out the same
    whichever part it is planned in, failing and capped queries included."""
    occ = W.plan_map(1024, 0.012)
    m = api.ObstacleMap(ctx, raster=occ)
    cs, cg = W.plan_queries(7000, 1024, 11, min_dist=20.0, max_dist=120.0)
    fs = api.collision_flags(ctx, m, CONF, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(ctx, m, CONF, cg[:, 0], cg[:, 1], cg[:, 2])
    inside = (cs[:, 0] > 8) & (cs[:, 0] < 1015) & (cs[:, 1] > 8) & (cs[:, 1] < 1015)
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:2100]
    assert len(keep) == 2100
    starts, goals = cs[keep].copy(), cg[keep].copy()
    starts[7] = goals[7]
    keys = ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")
    ctx.set_max_expansions(120)
    runs = []
    try:
        for parts in (1, 0, 3):
            ctx.set_batch_parts(parts)
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=1024, closed_cap=120)
            runs.append({k: np.array(r[k], copy=True) for k in keys})
            runs[-1]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
            runs[-1]["counters"] = api.plan_batch_counters(ctx)
    finally:
        ctx.set_batch_parts(0)
        ctx.set_max_expansions(100000)
    seen = set(int(v) for v in runs[0]["status"])
    assert {A.PP_OK, A.PP_ERR_LIMIT, A.PP_ERR_DEGENERATE_RS} <= seen, seen
    for which, r in enumerate(runs[1:], 1):
        for k in ("exp", "status") + keys:
            if not np.array_equal(runs[0][k], r[k], equal_nan=True):
                a0, a1 = runs[0][k].reshape(len(starts), -1), r[k].reshape(len(starts), -1)
                rows = np.nonzero(((a0 != a1) & ~((a0 != a0) & (a1 != a1))).any(axis=1))[0]
                i = int(rows[0])
                raise AssertionError(
                    f"run {which} differs from the unsplit run in {k!r} for {len(rows)} queries, first {i}: "
                    f"status {runs[0]['status'][i]} / {r['status'][i]}, closed {runs[0]['exp'][i]} / {r['exp'][i]}, "
                    f"cost {runs[0]['cost'][i]} / {r['cost'][i]}, len {runs[0]['len'][i]} / {r['len'][i]}, "
                    f"first differing closed position "
                    f"{int(np.argmax(runs[0]['closed'][i] != r['closed'][i])) if (runs[0]['closed'][i] != r['closed'][i]).any() else None}")
    # the automatic choice split this batch (more lanes than one part has), and all of it was planned
    assert runs[1]["counters"]["lanes"] > runs[0]["counters"]["lanes"]
    assert all(r["counters"]["queries"] == len(starts) for r in runs)


def test_host_thread_count_does_not_change_a_batch(ctx):
    """pp_ctx_set_host_threads only says how many threads apply a round's results."""
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    rng = np.random.default_rng(77)
    n = 300
    starts = np.column_stack([rng.uniform(4, 46, n), rng.uniform(4, 26, n), rng.uniform(-3, 3, n)])
    goals = np.column_stack([rng.uniform(4, 46, n), rng.uniform(4, 26, n), rng.uniform(-3, 3, n)])
    ctx.set_max_expansions(150)
    runs = []
    try:
        for threads in (15, 0, 3):
            ctx.set_host_threads(threads)
            r = api.plan_batch(ctx, m, CONF, starts, goals, cap=1024, closed_cap=150)
            runs.append({k: np.array(r[k], copy=True) for k in ("status", "len", "cost", "x", "y", "yaw", "dir", "closed")})
            runs[-1]["exp"] = np.array([int(s.expansions) for s in r["stats"]])
    finally:
        ctx.set_host_threads(15)
        ctx.set_max_expansions(100000)
    assert (runs[0]["status"] == 0).sum() > 20
    for r in runs[1:]:
        for k, v in runs[0].items():
            assert np.array_equal(v, r[k], equal_nan=True), k


@pytest.mark.parametrize("xy,yaw_deg,step,n_steer,check,extend,maxs,hc", [
    (1.0, 22.5, 0.2, 20, 3, 1.4, 0.6, 15.0),
    (3.0, 5.0, 0.7, 10, 8, 0.8, 0.8, 5.0),
    (1.5, 10.0, 0.3, 4, 1, 0.3, 0.4, 1.0),
    (2.0, 15.0, 0.4, 33, 5, 1.4, 0.6, 15.0),
])
def test_other_configs_vs_oracle(ctx, xy, yaw_deg, step, n_steer, check, extend, maxs, hc):
    """Rollouts, analytic shots and a planner batch under Configs other than the demo's: grid and
    yaw resolution, step size (hence poses per primitive), steer set, check step, bound, steering
    limit and heuristic weight all differ from src/main.rs:216-230."""
    conf = A.demo_config()
    conf.xy_reso, conf.yaw_reso, conf.move_step = xy, math.radians(yaw_deg), step
    conf.n_steer, conf.collision_check_step, conf.extend_bound = n_steer, check, extend
    conf.max_steer, conf.h_cost, conf.gear_cost, conf.steer_change_cost = maxs, hc, 10.0, 0.0
    rng = np.random.default_rng(int(xy * 1000 + n_steer))
    msz = 140
    occ = W.occupancy_grid(msz, 0.004, seed=4242 + n_steer)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    n = 300
    px, py = rng.uniform(3, msz - 4, n), rng.uniform(3, msz - 4, n)
    pyaw = rng.uniform(-2 * math.pi, 2 * math.pi, n)
    pdir = rng.choice([-1, 1], n)
    steer, _ = api.calc_motion_set(conf)
    assert len(steer) == 2 * (2 * n_steer - 1)
    psteer = rng.choice(steer, n)
    pcost = rng.uniform(0, 500, n)
    got = api.expand(ctx, m, conf, px, py, pyaw, pdir, psteer, pcost)
    st, exp = om.expand(conf, px, py, pyaw, pdir, psteer, pcost)
    assert st == 0
    amb = (got["valid"] & 2) != 0
    assert amb.sum() <= 2
    ok = ~amb
    assert ((got["valid"] & 1)[ok] == exp["valid"][ok]).all()
    v = ok & (exp["valid"] == 1)
    assert v.sum() > 500
    for k in ("xind", "yind", "yawind"):
        assert (got[k][v] == exp[k][v]).all(), k
    assert (got["cost"][v] == exp["cost"][v]).all() and (got["end_yaw"][v] == exp["end_yaw"][v]).all()
    ns = 200
    s = np.c_[rng.uniform(3, msz - 4, ns), rng.uniform(3, msz - 4, ns), rng.uniform(-math.pi, math.pi, ns)]
    g = s + rng.uniform(-25, 25, (ns, 3)) * [1, 1, 0.1]
    found, idx, npaths, words, cost = api.analytic_expansion(ctx, m, conf, s, g)
    st, efound, eidx, enp, ewords, ecost = om.analytic(conf, s, g)
    oks = (found & 2) == 0
    assert oks.mean() > 0.9
    assert ((found & 1)[oks] == (efound & 1)[oks]).all() and (idx[oks] == eidx[oks]).all()
    assert (npaths[oks] == enp[oks]).all() and np.allclose(cost[oks], ecost[oks], rtol=1e-9, atol=0)
    nq = 24
    starts = np.c_[rng.uniform(12, msz - 12, nq), rng.uniform(12, msz - 12, nq), rng.uniform(-3, 3, nq)]
    goals = np.c_[rng.uniform(15, msz - 15, 3), rng.uniform(15, msz - 15, 3), rng.uniform(-3, 3, 3)][rng.integers(0, 3, nq)]
    ctx.set_max_expansions(800)
    O.set_max_expansions(800)
    try:
        res = api.plan_batch(ctx, m, conf, starts, goals, cap=40000, closed_cap=1 << 12)
        differ = found_n = 0
        for i in range(nq):
            e = om.plan(conf, starts[i], goals[i])
            differ += not same_batch_query(res, i, e, 1 << 12)
            found_n += res["status"][i] == 0
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    assert differ == 0 and found_n >= nq // 3, (differ, found_n)
