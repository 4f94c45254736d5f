"""Parity of the obstacle-raster and wavefront kernels against the CPU oracle (bit-exact)."""
import ctypes as C
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


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def same_grid(got, exp):
    assert got.shape == exp.shape
    assert (np.isinf(got) == np.isinf(exp)).all()
    fin = np.isfinite(exp)
    assert (got[fin] == exp[fin]).all()  # bit-exact


def test_heuristic_golden(ctx):
    ox, oy = W.demo_obstacles()
    for h in json.load(open(os.path.join(G, "heuristic.json"))):
        hox, hoy = (h["ox"], h["oy"]) if "ox" in h else (ox, oy)
        m = api.ObstacleMap(ctx, hox, hoy)
        obs, dims = api.calc_obsmap(ctx, m, h["reso"], h["rr"])
        assert dims.tolist() == h["dims"] and obs.tolist() == h["obsmap"]
        hm, dims = api.calc_holonomic_heuristic_with_obstacle(ctx, m, h["gx"], h["gy"], h["reso"], h["rr"])
        exp = np.array([[math.inf if v == "inf" else v for v in row] for row in h["hmap"]])
        same_grid(hm, exp)


def test_heuristic_demo_goal_rust_rounding(ctx):
    # 45/2 = 22.5 rounds half away from zero to cell 23 (the Python twin would pick 22)
    ox, oy = W.demo_obstacles()
    m = api.ObstacleMap(ctx, ox, oy)
    hm, dims = api.calc_holonomic_heuristic_with_obstacle(ctx, m, 45.0, 20.0, 2.0, 1.0)
    st, exp, edims = O.heuristic(ox, oy, 45.0, 20.0, 2.0, 1.0)
    assert st == 0 and dims.tolist() == edims.tolist()
    same_grid(hm, exp)
    assert hm[23, 10] == 0.0 and np.isfinite(hm).sum() == 302


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_obsmap_and_heuristic_random_points_vs_oracle(ctx, seed):
    rng = np.random.default_rng(seed)
    n = 600
    ox, oy = rng.uniform(0, 90, n), rng.uniform(0, 70, n)   # non-lattice obstacle points
    ox[:4], oy[:4] = [0, 90, 0, 90], [0, 0, 70, 70]
    reso, rr = [(1.0, 0.7), (2.0, 1.0), (0.5, 1.3)][seed - 1]
    m = api.ObstacleMap(ctx, ox, oy)
    obs, dims = api.calc_obsmap(ctx, m, reso, rr)
    st, eobs, edims = O.obsmap(ox, oy, reso, rr, 1)
    assert dims.tolist() == edims.tolist() and (obs == eobs).all()
    gx, gy = 45.3, 33.1
    hm, _ = api.calc_holonomic_heuristic_with_obstacle(ctx, m, gx, gy, reso, rr)
    st, exp, _ = O.heuristic(ox, oy, gx, gy, reso, rr)
    assert st == 0
    same_grid(hm, exp)


def test_obsmap_exact_threshold_cases(ctx):
    """Distances that land exactly on (or one ulp around) rr/reso: the decision must be the
    one a correctly rounded hypot gives (glibc's on the host)."""
    pts = [(10.0, 10.0), (13.0, 14.0), (20.5, 10.0), (30.0, 30.0)]
    ox = np.array([0.0, 40.0] + [p[0] for p in pts])
    oy = np.array([0.0, 40.0] + [p[1] for p in pts])
    m = api.ObstacleMap(ctx, ox, oy)
    for rr in (5.0, np.nextafter(5.0, 0), np.nextafter(5.0, 9), math.hypot(3.0, 4.0), 0.5, 2.5,
               math.hypot(1.0, 2.0), np.nextafter(math.hypot(1.0, 2.0), 0), math.sqrt(2.0),
               np.nextafter(math.sqrt(2.0), 0), np.nextafter(math.sqrt(2.0), 9)):
        obs, dims = api.calc_obsmap(ctx, m, 1.0, float(rr))
        st, eobs, _ = O.obsmap(ox, oy, 1.0, float(rr), 0)
        assert (obs == eobs).all(), rr


@pytest.mark.parametrize("n,p,kind", [(96, 0.2, "bernoulli"), (300, 0.3, "bernoulli"),
                                      (257, 0.0, "walls"), (512, 0.15, "bernoulli")])
def test_wavefront_vs_oracle(ctx, n, p, kind):
    import torch
    obs, gx, gy = W.wavefront_obsmap(n, p, kind=kind)
    d_obs = torch.from_numpy(obs).cuda()
    cost = torch.empty((n, n), dtype=torch.float64, device="cuda")
    iters = api.wavefront_dev(ctx, d_obs, gx, gy, cost)
    ctx.synchronize()
    st, exp = O.wavefront(obs, gx, gy)
    assert st == 0
    same_grid(cost.cpu().numpy(), exp)
    assert iters > 0


@pytest.mark.parametrize("xw,yw", [(1, 1), (1, 7), (5, 1), (2, 2), (33, 65), (65, 33), (130, 67), (3, 200),
                                   (64, 64), (63, 66), (129, 128)])
def test_wavefront_ragged_shapes_vs_oracle(ctx, xw, yw):
    """Grids smaller than a tile, sides that are not multiples of the 4-cell patch or the 32-cell
    block, odd row lengths (the dense arrays' rows are then not 16-byte aligned), and a view that
    starts 8 bytes into an allocation."""
    import torch
    u = W.uniform01(1234 + xw * 1000 + yw, 0, xw * yw).reshape(xw, yw)
    obs = (u < 0.2).astype(np.uint8)
    gx, gy = xw // 2, yw // 2
    obs[gx, gy] = 0
    d_obs = torch.from_numpy(obs).cuda()
    st, exp = O.wavefront(obs, gx, gy)
    assert st == 0
    cost = torch.empty((xw, yw), dtype=torch.float64, device="cuda")
    api.wavefront_dev(ctx, d_obs, gx, gy, cost)
    ctx.synchronize()
    same_grid(cost.cpu().numpy(), exp)
    # the same through misaligned views: costs 8 bytes, raster 1 byte into their allocations
    flat = torch.empty(xw * yw + 1, dtype=torch.float64, device="cuda")
    oflat = torch.empty(xw * yw + 1, dtype=torch.uint8, device="cuda")
    c2, o2 = flat[1:].view(xw, yw), oflat[1:].view(xw, yw)
    o2.copy_(d_obs)
    torch.cuda.synchronize()  # the library works on its own stream
    api.wavefront_dev(ctx, o2, gx, gy, c2)
    ctx.synchronize()
    same_grid(c2.cpu().numpy(), exp)


def test_wavefront_goal_on_obstacle_and_border_and_errors(ctx):
    import torch
    obs, gx, gy = W.wavefront_obsmap(64, 0.2)
    obs[10, 10] = 1
    for (x, y) in [(10, 10), (0, 5), (63, 63)]:
        d_obs = torch.from_numpy(obs).cuda()
        cost = torch.empty((64, 64), dtype=torch.float64, device="cuda")
        api.wavefront_dev(ctx, d_obs, x, y, cost)
        ctx.synchronize()
        st, exp = O.wavefront(obs, x, y)
        same_grid(cost.cpu().numpy(), exp)
    with pytest.raises(api.PlannerError) as e:
        api.wavefront_dev(ctx, d_obs, 64, 3, cost)
    assert e.value.status == A.PP_ERR_BOUNDS


def test_wavefront_rectangular_and_fixpoint_property_large(ctx):
    """2048 x 1024 grid: too slow for the oracle in a unit test, so check the defining property
    instead -- the result is a fixpoint of the relaxation and 0 at the goal."""
    import torch
    xw, yw = 2048, 1024
    u = W.uniform01(77, 0, xw * yw).reshape(xw, yw)
    obs = (u < 0.25).astype(np.uint8)
    obs[0, :] = obs[-1, :] = 1
    obs[:, 0] = obs[:, -1] = 1
    gx, gy = 1000, 500
    obs[gx, gy] = 0
    d_obs = torch.from_numpy(obs).cuda()
    cost = torch.empty((xw, yw), dtype=torch.float64, device="cuda")
    api.wavefront_dev(ctx, d_obs, gx, gy, cost)
    ctx.synchronize()
    d = cost
    assert float(d[gx, gy]) == 0.0
    inf = float("inf")
    valid = torch.zeros((xw, yw), dtype=torch.bool, device="cuda")
    valid[1:, 1:] = True
    valid &= d_obs == 0
    p = torch.nn.functional.pad(d, (1, 1, 1, 1), value=inf)
    best = torch.full_like(d, inf)
    s2 = math.hypot(1.0, 1.0)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            w = 1.0 if dx == 0 or dy == 0 else s2
            best = torch.minimum(best, p[1 + dx:1 + dx + xw, 1 + dy:1 + dy + yw] + w)
    best = torch.where(valid, best, torch.full_like(best, inf))
    best[gx, gy] = 0.0
    assert bool(((best == d) | (torch.isinf(best) & torch.isinf(d))).all())
    assert int(torch.isfinite(d).sum()) > xw * yw // 2


def _fixpoint_violations(d, d_obs, gx, gy):
    """Cells where d is not min over neighbours of (d[n] + w), evaluated on the device."""
    import torch
    xw, yw = d.shape
    inf = float("inf")
    p = torch.nn.functional.pad(d, (1, 1, 1, 1), value=inf)
    best = torch.full_like(d, inf)
    s2 = math.hypot(1.0, 1.0)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            w = 1.0 if dx == 0 or dy == 0 else s2
            best = torch.minimum(best, p[1 + dx:1 + dx + xw, 1 + dy:1 + dy + yw] + w)
    del p
    best[0, :] = inf
    best[:, 0] = inf
    best = torch.where(d_obs == 0, best, torch.full_like(best, inf))
    best[gx, gy] = 0.0
    bad = ~((best == d) | (torch.isinf(best) & torch.isinf(d)))
    return int(bad.sum())


def test_wavefront_full_size_16384_fixpoint(ctx):
    """The 16384 x 16384 benchmark grid: result is the relaxation's fixpoint, 0 at the goal, and
    most of the free space is reached."""
    import torch
    n = 16384
    obs, gx, gy = W.wavefront_obsmap(n, 0.2)
    d_obs = torch.from_numpy(obs).cuda()
    cost = torch.empty((n, n), dtype=torch.float64, device="cuda")
    api.wavefront_dev(ctx, d_obs, gx, gy, cost)
    ctx.synchronize()
    assert float(cost[gx, gy]) == 0.0
    assert _fixpoint_violations(cost, d_obs, gx, gy) == 0
    assert int(torch.isfinite(cost).sum()) > 0.7 * n * n


def test_relax_band_does_not_change_the_result(ctx):
    """The scheduling knob only reorders the relaxation."""
    import torch
    obs, gx, gy = W.wavefront_obsmap(700, 0.25, seed=123)
    d_obs = torch.from_numpy(obs).cuda()
    outs = []
    for band in (1.0, 64.0, 1.0e9):
        assert ctx.lib.pp_ctx_set_relax_band(ctx.h, band) == 0
        cost = torch.empty((700, 700), dtype=torch.float64, device="cuda")
        api.wavefront_dev(ctx, d_obs, gx, gy, cost)
        ctx.synchronize()
        outs.append(cost.cpu().numpy())
    assert ctx.lib.pp_ctx_set_relax_band(ctx.h, 128.0) == 0
    st, exp = O.wavefront(obs, gx, gy)
    for o in outs:
        assert (o.view(np.uint64) == exp.view(np.uint64)).all()


def test_whole_grid_session_equals_the_dense_entry_point(ctx):
    """A slab session over all columns keeps the grid in the relaxation's own layout (raster packed
    once, nothing unpacked until asked): what it reads out equals pp_wavefront_dev, also after a
    reset onto another raster."""
    import torch
    from path_planner_amd.tiled import DeviceSlabSolver, make_slab
    n = 300
    sess = None
    for seed in (5, 6):
        obs, gx, gy = W.wavefront_obsmap(n, 0.25, seed=seed)
        d_obs = torch.from_numpy(obs).cuda()
        cost = torch.empty((n, n), dtype=torch.float64, device="cuda")
        api.wavefront_dev(ctx, d_obs, gx, gy, cost)
        ctx.synchronize()
        slab_obs, _ = make_slab(obs, n, n, 0, n, "cuda")
        torch.cuda.synchronize()
        if sess is None:
            sess = DeviceSlabSolver(ctx, n, n, 0, n, slab_obs, (gx, gy))
        else:
            assert (gx, gy) == goal  # the session keeps its goal
            sess.reset(slab_obs)
        goal = (gx, gy)
        assert sess.step(None, None, float("inf")) == float("inf")
        out = sess.read()
        torch.cuda.synchronize()
        assert torch.equal(out[1:-1], cost)
        assert bool(torch.isinf(out[0]).all()) and bool(torch.isinf(out[-1]).all())
        # single cells straight from the working copy
        g = torch.Generator(device="cpu").manual_seed(seed)
        xs = torch.randint(0, n, (5000,), generator=g).cuda()
        ys = torch.randint(0, n, (5000,), generator=g).cuda()
        assert torch.equal(sess.gather(xs, ys), cost[xs, ys])
    with pytest.raises(api.PlannerError) as e:
        sess.gather(torch.tensor([n + 1], device="cuda"), torch.tensor([0], device="cuda"))
    assert e.value.status == A.PP_ERR_BOUNDS
    sess.close()


@pytest.mark.parametrize("xw,yw,nctx,band", [(300, 300, 2, 0.0), (300, 300, 3, 16.0), (257, 130, 3, 1.0),
                                              (5, 40, 5, 0.0), (700, 512, 4, 64.0), (64, 64, 1, 0.0)])
def test_wavefront_tiled_over_contexts_equals_one_relaxation(ctx, xw, yw, nctx, band):
    """pp_wavefront_tiled cuts the raster into column slabs, one per context (here: several
    contexts on the one GPU, as the devices of a node would be), relaxes them in cost order with
    their boundary columns copied across, and gives the costs of one relaxation of the whole
    raster -- the oracle's, bit for bit -- whatever the number of slabs and the band."""
    u = W.uniform01(77 + xw * 1000 + yw, 0, xw * yw).reshape(xw, yw)
    obs = (u < 0.25).astype(np.uint8)
    gx, gy = (3 * xw) // 4, yw // 3  # off-centre: the goal's slab is not the middle one
    obs[gx, gy] = 0
    ctxs = [ctx] + [api.Context(0) for _ in range(nctx - 1)]
    try:
        cost, rounds = api.wavefront_tiled(ctxs, obs, gx, gy, band)
        st, exp = O.wavefront(obs, gx, gy)
        assert st == 0
        same_grid(cost, exp)
        assert rounds >= 1
        if nctx > 1 and band and band <= 16.0 and xw >= 257:
            assert rounds > 3, rounds  # a narrow band takes many exchanges to the same result
        # errors: the same context twice, more slabs than columns, a goal outside
        if nctx > 1:
            with pytest.raises(api.PlannerError):
                api.wavefront_tiled([ctx, ctx], obs, gx, gy)
        with pytest.raises(api.PlannerError):
            api.wavefront_tiled(ctxs, obs[:max(1, nctx - 1)], 0, 0) if nctx > 1 else api.wavefront_tiled(ctxs, obs, xw, 0)
    finally:
        for c in ctxs[1:]:
            c.close()


def test_slab_outlives_its_context():
    """A context destroyed before its slab session takes the slab's device memory with it: the
    handle then refuses every call and pp_slab_destroy still works (two sessions, one destroyed in
    the right order first, so the context's list has been through an unlink)."""
    import torch
    from path_planner_amd.tiled import DeviceSlabSolver, make_slab
    n = 200
    own = api.Context(0)
    obs, gx, gy = W.wavefront_obsmap(n, 0.2, seed=9)
    slab_obs, _ = make_slab(obs, n, n, 0, n, "cuda")
    torch.cuda.synchronize()
    a = DeviceSlabSolver(own, n, n, 0, n, slab_obs, (gx, gy))
    b = DeviceSlabSolver(own, n, n, 0, n, slab_obs, (gx, gy))
    c = DeviceSlabSolver(own, n, n, 0, n, slab_obs, (gx, gy))
    assert b.step(None, None, float("inf")) == float("inf")
    b.close()
    lib = own.lib
    own.close()
    for s in (a, c):
        assert lib.pp_slab_relax(s.h, None, None, C.c_double(1.0), None, None, None, None) == A.PP_ERR_INVALID
        assert lib.pp_slab_reset(s.h, slab_obs.data_ptr()) == A.PP_ERR_INVALID
        lib.pp_slab_destroy(s.h)
        s.h = None
