"""Parity of the HIP footprint-collision and motion-primitive kernels against the CPU oracle."""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import api
from path_planner_amd import workloads as W

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CONF = A.demo_config()


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def assert_flags_equal(got, exp):
    """bit0 must match the oracle wherever the kernel did not raise its guard-band bit."""
    hit = got & 1
    amb = (got & 2) != 0
    assert not (got & 4).any()
    bad = (hit != exp) & ~amb
    assert not bad.any(), f"{bad.sum()} flag mismatches outside the guard band"
    assert amb.sum() <= max(2, len(got) // 100000), f"{amb.sum()} guard-band poses"
    return int((hit != exp).sum())


def test_collision_golden_demo_and_scatter(ctx):
    col = json.load(open(os.path.join(G, "collision.json")))
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    assert m.is_lattice and m.num_obstacles == len(ox)
    poses = np.array(col["demo"]["poses"])
    got = api.collision_flags(ctx, m, CONF, poses[:, 0], poses[:, 1], poses[:, 2])
    assert assert_flags_equal(got, np.array(col["demo"]["flags"], np.uint8)) == 0
    sc = col["scatter"]
    m2 = api.ObstacleMap(ctx, sc["ox"], sc["oy"])
    assert not m2.is_lattice
    poses = np.array(sc["poses"])
    got = api.collision_flags(ctx, m2, CONF, poses[:, 0], poses[:, 1], poses[:, 2])
    assert assert_flags_equal(got, np.array(sc["flags"], np.uint8)) == 0


@pytest.mark.parametrize("size,p,n", [(256, 0.05, 200000), (512, 0.01, 200000), (192, 0.3, 50000)])
def test_collision_random_grid_vs_oracle(ctx, size, p, n):
    occ = W.occupancy_grid(size, p)
    ox, oy = W.raster_to_points(occ)
    x, y, yaw = W.random_poses(n, size)
    # include poses hanging over the border and far outside the map
    x[:100] -= 30.0
    y[100:200] += size
    m = api.ObstacleMap(ctx, raster=occ)
    assert m.is_lattice
    got = api.collision_flags(ctx, m, CONF, x, y, yaw)
    st, exp = O.OracleMap(ox, oy).collision_flags(CONF, x, y, yaw)
    assert st == 0
    assert assert_flags_equal(got, exp) == 0
    assert 0.02 < exp.mean() < 1.0
    # same obstacle set as a point list (bucket path forced by a tiny non-lattice offset copy)
    m2 = api.ObstacleMap(ctx, ox, oy)
    got2 = api.collision_flags(ctx, m2, CONF, x, y, yaw)
    assert (got2 == got).all()


def small_car_config(front, back, width, extend):
    c = A.demo_config()
    c.car.center2front, c.car.center2back, c.car.width = front, back, width
    c.extend_bound = extend
    return c


@pytest.mark.parametrize("front,back,width,extend", [
    (3.0, 1.0, 1.8, 0.2),                    # a compact car: bounding radius 2.46 cells, eight-row window
    (1.0, 1.0, 1.0, 0.0),                    # radius 1.12
    (4.0, 1.0, 2.0, 0.2499),                 # radius 2.9995: just inside the eight-row window's limit
    (4.0, 1.0, 2.0, 0.2509),                 # radius 3.0005: just outside, sixteen rows
])
def test_collision_small_footprints_vs_oracle(ctx, front, back, width, extend):
    """Footprints whose bounding radius is below three cells take the filter's eight-row window;
    flags must equal the oracle's on either side of that limit."""
    conf = small_car_config(front, back, width, extend)
    for size, p, n in ((256, 0.05, 150000), (192, 0.3, 50000)):
        occ = W.occupancy_grid(size, p)
        ox, oy = W.raster_to_points(occ)
        x, y, yaw = W.random_poses(n, size)
        x[:100] -= 30.0
        y[100:200] += size
        yaw[200:264] = np.repeat([0.0, math.pi / 2, -math.pi / 2, math.pi], 16)  # axis-aligned edges
        m = api.ObstacleMap(ctx, raster=occ)
        got = api.collision_flags(ctx, m, conf, x, y, yaw)
        st, exp = O.OracleMap(ox, oy).collision_flags(conf, x, y, yaw)
        assert st == 0
        assert assert_flags_equal(got, exp) == 0
        assert 0.01 < exp.mean() < 1.0


def test_expand_small_footprint_vs_oracle(ctx):
    """The rollout kernel's subsampled collision checks with the eight-row window."""
    conf = small_car_config(3.0, 1.0, 1.8, 0.2)
    occ = W.occupancy_grid(128, 0.03)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    rng = np.random.default_rng(21)
    n = 400
    px, py = rng.uniform(3, 124, n), rng.uniform(3, 124, n)
    pyaw = rng.uniform(-math.pi, math.pi, n)
    pdir = rng.choice([-1, 1], n)
    steer, _ = api.calc_motion_set(conf)
    psteer = rng.choice(steer, n)
    pcost = rng.uniform(0, 100, n)
    got = api.expand(ctx, m, conf, px, py, pyaw, pdir, psteer, pcost)
    st, exp = om.expand(conf, px, py, pyaw, pdir, psteer, pcost)
    assert st == 0
    ok = (got["valid"] & 2) == 0
    assert (~ok).sum() <= 2
    assert ((got["valid"] & 1)[ok] == exp["valid"][ok]).all()
    assert 0.05 < exp["valid"].mean() < 0.95


def test_collision_bucket_path_vs_oracle(ctx):
    rng = np.random.default_rng(5)
    ox, oy = rng.uniform(-20, 80, 3000), rng.uniform(-20, 80, 3000)
    x, y, yaw = rng.uniform(-30, 90, 100000), rng.uniform(-30, 90, 100000), rng.uniform(-9, 9, 100000)
    m = api.ObstacleMap(ctx, ox, oy)
    assert not m.is_lattice
    got = api.collision_flags(ctx, m, CONF, x, y, yaw)
    st, exp = O.OracleMap(ox, oy).collision_flags(CONF, x, y, yaw)
    assert assert_flags_equal(got, exp) == 0


def test_collision_edge_cases(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    # empty batch
    assert len(api.collision_flags(ctx, m, CONF, [], [], [])) == 0
    # non-finite poses are reported, not crashed on
    got = api.collision_flags(ctx, m, CONF, [float("nan"), 25.0, float("inf")], [5.0, 8.0, 1.0],
                              [0.0, float("nan"), 0.0])
    assert (got & 4).all()
    with pytest.raises(api.PlannerError) as e:
        api.is_collision(ctx, m, CONF, [float("nan")], [0.0], [0.0])
    assert e.value.status == A.PP_ERR_NONFINITE
    # list form: any-of
    assert api.is_collision(ctx, m, CONF, [25.0, 20.0], [8.0, 8.0], [0.3, 0.3]) is True
    assert api.is_collision(ctx, m, CONF, [25.0], [8.0], [0.3]) is False
    # exact lattice-aligned pose: rectangle edge exactly on an obstacle column (strict <)
    # car centre at x = 1.75 + 2.4 = 4.15 from obstacle x=0 => |dx| == bound_len exactly -> no hit
    got = api.collision_flags(ctx, m, CONF, [4.15 - 1.75], [15.0], [0.0])
    st, exp = O.OracleMap(ox, oy).collision_flags(CONF, [4.15 - 1.75], [15.0], [0.0])
    assert (got[0] & 1) == exp[0] or (got[0] & 2)


def test_collision_invariance_properties_full_size(ctx):
    """Size-independent checks at the benchmark scale (2048^2 grid, 4M poses): the flags do not
    depend on how the obstacle set is presented, and translating map and poses together by an
    integer vector leaves them unchanged."""
    import torch
    size, n = 2048, 1 << 22
    occ = W.occupancy_grid(size, 0.05)
    x, y, yaw = W.random_poses(n, size)
    m = api.ObstacleMap(ctx, raster=occ)
    dx, dy, dyaw = (torch.from_numpy(a).cuda() for a in (x, y, yaw))
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    api.collision_flags_dev(ctx, m, CONF, dx, dy, dyaw, flags)
    ctx.synchronize()
    f0 = flags.cpu().numpy()
    m2 = api.ObstacleMap(ctx, raster=occ, origin=(-1000, 4096))
    dx2, dy2 = dx - 1000.0, dy + 4096.0
    api.collision_flags_dev(ctx, m2, CONF, dx2, dy2, dyaw, flags)
    ctx.synchronize()
    f1 = flags.cpu().numpy()
    # translation by integers is exact for |coords| < 2^53 but the footprint centre rounds
    # differently; only guard-band poses may differ
    diff = (f0 & 1) != (f1 & 1)
    assert diff.sum() <= 2
    assert 0.85 < (f0 & 1).mean() < 0.97
    # oracle on a 200k-pose sample of the same batch
    ox, oy = W.raster_to_points(occ)
    st, exp = O.OracleMap(ox, oy).collision_flags(CONF, x[:200000], y[:200000], yaw[:200000], threads=8)
    assert assert_flags_equal(f0[:200000], exp) == 0


def test_expand_vs_oracle(ctx):
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    om = O.OracleMap(ox, oy)
    rng = np.random.default_rng(9)
    n = 300
    px, py = rng.uniform(2, 48, n), rng.uniform(2, 28, n)
    pyaw = rng.uniform(-2 * math.pi, 2 * math.pi, n)
    pdir = rng.choice([-1, 1], n)
    steer, _ = api.calc_motion_set(CONF)
    psteer = rng.choice(steer, n)
    pcost = rng.uniform(0, 500, n)
    px[0], py[0], pyaw[0], pdir[0], psteer[0], pcost[0] = 10.0, 7.0, math.radians(120.0), 1, 0.0, 0.0
    got = api.expand(ctx, m, CONF, px, py, pyaw, pdir, psteer, pcost)
    st, exp = om.expand(CONF, px, py, pyaw, pdir, psteer, pcost)
    assert st == 0
    amb = (got["valid"] & 2) != 0
    assert amb.sum() <= 2
    ok = ~amb
    assert ((got["valid"] & 1)[ok] == exp["valid"][ok]).all()
    v = ok & (exp["valid"] == 1)
    assert v.sum() > 1000
    for k in ("xind", "yind", "yawind"):
        assert (got[k][v] == exp[k][v]).all(), k
    assert (got["cost"][v] == exp["cost"][v]).all()          # trig-free: bit-exact
    assert (got["end_yaw"][v] == exp["end_yaw"][v]).all()    # tan(u) table: bit-exact
    assert np.allclose(got["end_x"][v], exp["end_x"][v], rtol=1e-12, atol=1e-12)
    assert np.allclose(got["end_y"][v], exp["end_y"][v], rtol=1e-12, atol=1e-12)


def test_expand_vs_unpatched_twin_fixture(ctx):
    """tests/golden/twin_next_node.json: calc_next_node of the reference's Python twin on inputs
    where it equals the crate (no yaw wrap, no rounding tie)."""
    from test_oracle_golden import check_next_node_fixture
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    out = check_next_node_fixture(lambda *a: api.expand(ctx, m, CONF, *a))
    assert ((out["valid"] & 2) != 0).sum() <= 2


def test_rollout_points_vs_oracle(ctx):
    steer, direc = api.calc_motion_set(CONF)
    for i in (0, 19, 25, 50, 77):
        x, y, yaw = api.rollout_points(ctx, CONF, 10.0, 7.0, 2.0, steer[i], direc[i])
        ex, ey, eyaw = O.rollout_points(CONF, 10.0, 7.0, 2.0, steer[i], direc[i])
        assert len(x) == 11
        assert (yaw == eyaw).all()
        assert np.allclose(x, ex, rtol=0, atol=1e-13) and np.allclose(y, ey, rtol=0, atol=1e-13)


def test_expand_dev_matches_host_form(ctx):
    """pp_expand_batch_dev (device pointers) against pp_expand_batch (host pointers)."""
    import torch
    occ = W.occupancy_grid(256, 0.02, seed=17)
    m = api.ObstacleMap(ctx, raster=occ)
    n = 3000
    x, y, yaw = W.random_poses(n, 256, seed=91)
    pdir = np.where(np.arange(n) % 3 == 0, -1, 1).astype(np.int64)
    ps = np.linspace(-0.5, 0.5, n)
    pc = np.linspace(0.0, 100.0, n)
    host = api.expand(ctx, m, CONF, x, y, yaw, pdir, ps, pc)
    P = len(host["valid"]) // n
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = {k: torch.zeros(n * P, dtype=(torch.uint8 if k == "valid" else torch.int64 if k in
                                        ("xind", "yind", "yawind") else torch.float64), device="cuda")
           for k in ("valid", "xind", "yind", "yawind", "cost", "end_x", "end_y", "end_yaw")}
    api.expand_dev(ctx, m, CONF, t(x), t(y), t(yaw), t(pdir), t(ps), t(pc), out)
    ctx.synchronize()
    for k, v in out.items():
        got = v.cpu().numpy()
        exp = host[k]
        if exp.dtype == np.uint64:
            exp = exp.view(np.int64)
        assert (got == exp).all(), k
