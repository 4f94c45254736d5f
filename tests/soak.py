"""Parity soak on seeds the test suite does not use (GPU box): the collision, Reeds-Shepp, heuristic and
planner paths against the CPU oracle with the tests' own checkers.  usage: tests/soak.py [rounds]
(not collected by pytest: a longer run of the same checks, started by hand)
Prints one line per item and a summary; exits non-zero at the first difference a guard band does
not account for."""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from path_planner_amd import _abi as A  # noqa: E402
from path_planner_amd import api  # noqa: E402
from path_planner_amd import workloads as W  # noqa: E402
import test_gpu_collision as TC  # noqa: E402
import test_gpu_planner as TP  # noqa: E402
import test_gpu_rs as TR  # noqa: E402

CONF = A.demo_config()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.init()  # before the library's own first HIP call
ctx = api.Context(0)
t_start = time.time()


def say(*a):
    print(f"[{time.time() - t_start:6.1f}s]", *a, flush=True)


for r in range(rounds):
    seed = 0xA11CE000 + 977 * r
    rng = np.random.default_rng(seed)
    # ---- collision: random lattice maps (also off-origin), random and axis-aligned poses
    size = int(rng.integers(96, 700))
    p = float(rng.choice([0.004, 0.02, 0.05, 0.15, 0.35]))
    occ = W.occupancy_grid(size, p, seed=seed)
    x0, y0 = int(rng.integers(-500, 500)), int(rng.integers(-500, 500))
    ox, oy = W.raster_to_points(occ, x0, y0)
    n = 300000
    x, y, yaw = W.random_poses(n, size, seed=seed + 1)
    x += x0
    y += y0
    x[:200] -= 40.0
    y[200:400] += size
    yaw[400:464] = np.repeat([0.0, math.pi / 2, -math.pi / 2, math.pi], 16)
    yaw[464:528] *= 1.0e4
    conf = CONF if r % 2 == 0 else TC.small_car_config(float(rng.uniform(1, 4)), 1.0, float(rng.uniform(1, 2.4)),
                                                      float(rng.uniform(0, 0.5)))
    m = api.ObstacleMap(ctx, ox, oy)
    got = api.collision_flags(ctx, m, conf, x, y, yaw)
    st, exp = O.OracleMap(ox, oy).collision_flags(conf, x, y, yaw)
    assert st == 0
    TC.assert_flags_equal(got, exp)
    say(f"round {r}: collision {size}^2 p={p} origin ({x0},{y0}) lattice={m.is_lattice}: {n} poses equal, "
        f"{int(((got & 2) != 0).sum())} in the guard band, hit rate {exp.mean():.2f}")

    # ---- Reeds-Shepp batch: words, counts and selection
    npair = 150000
    q0, q1 = W.rs_pairs(npair, seed=seed + 2, extent=float(rng.choice([20.0, 100.0, 400.0])))
    maxc = W.demo_maxc() * float(rng.choice([0.5, 1.0, 2.0]))
    shares = TR.check_batch_against_oracle(ctx, q0, q1, maxc, 0.02, f"soak round {r}")
    say(f"round {r}: rs_batch {npair} pairs maxc {maxc:.4f}: clean pairs equal, tie-only pairs list the oracle's words, "
        f"shares {shares}")

    # ---- heuristic grid: ragged shapes, bitwise
    xw, yw = int(rng.integers(40, 1500)), int(rng.integers(40, 1500))
    obs = (rng.random((xw, yw)) < float(rng.choice([0.1, 0.2, 0.3, 0.38]))).astype(np.uint8)
    obs[0, :] = obs[-1, :] = 1
    obs[:, 0] = obs[:, -1] = 1
    gx, gy = int(rng.integers(1, xw - 1)), int(rng.integers(1, yw - 1))
    obs[gx, gy] = 0
    d_obs = torch.from_numpy(obs).cuda()
    d_cost = torch.empty((xw, yw), dtype=torch.float64, device="cuda")
    api.wavefront_dev(ctx, d_obs, gx, gy, d_cost)
    ctx.synchronize()
    st, ref = O.wavefront(obs, gx, gy)
    assert st == 0
    assert (d_cost.cpu().numpy().view(np.uint64) == ref.view(np.uint64)).all()
    # the same raster in column slabs over several contexts of this process
    nslab = int(rng.integers(2, 5))
    band = float(rng.choice([0.0, 4.0, 64.0, 700.0]))
    helpers = [api.Context(0) for _ in range(nslab - 1)]
    try:
        tcost, trounds = api.wavefront_tiled([ctx] + helpers, obs, gx, gy, band)
    finally:
        for h in helpers:
            h.close()
    assert (tcost.view(np.uint64) == ref.view(np.uint64)).all()
    say(f"round {r}: wavefront {xw}x{yw}: bitwise equal, {int(np.isfinite(ref).sum())} cells reached; "
        f"{nslab} slabs, band {band:g}: equal after {trounds} exchange rounds")

    # ---- planner: a batch on a random map against the oracle, closing order and path
    msz = int(rng.integers(90, 200))
    occ = W.occupancy_grid(msz, float(rng.choice([0.002, 0.004, 0.008])), seed=seed + 3)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    nq = 48
    goals_pool = np.c_[rng.uniform(15, msz - 15, 5), rng.uniform(15, msz - 15, 5), rng.uniform(-3, 3, 5)]
    starts = np.c_[rng.uniform(12, msz - 12, nq), rng.uniform(12, msz - 12, nq), rng.uniform(-3, 3, nq)]
    goals = goals_pool[rng.integers(0, 5, nq)]
    ctx.set_max_expansions(1200)
    O.set_max_expansions(1200)
    try:
        res = api.plan_batch(ctx, m, CONF, starts, goals, cap=20000, closed_cap=1 << 12)
        differ = found = 0
        for i in range(nq):
            e = om.plan(CONF, starts[i], goals[i])
            differ += not TP.same_batch_query(res, i, e, 1 << 12)
            found += res["status"][i] == 0
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    say(f"round {r}: planner {nq} queries on a {msz}^2 map: {found} found, {differ} differ after a guard-band decision")

# ---- other Configs than the demo's: resolutions, steer set, step sizes, cost weights
for r in range(rounds):
    seed = 0xC0F16000 + 613 * r
    rng = np.random.default_rng(seed)
    conf = A.demo_config()
    conf.xy_reso = float(rng.choice([1.0, 1.5, 2.0, 3.0]))
    conf.yaw_reso = math.radians(float(rng.choice([5.0, 10.0, 15.0, 22.5])))
    conf.move_step = float(rng.choice([0.2, 0.3, 0.4, 0.7]))
    conf.n_steer = int(rng.choice([4, 10, 20, 33]))
    conf.collision_check_step = int(rng.choice([1, 3, 5, 8]))
    conf.extend_bound = float(rng.choice([0.3, 0.8, 1.4]))
    conf.gear_cost = float(rng.choice([0.0, 10.0, 100.0]))
    conf.backward_cost = float(rng.choice([1.0, 5.0]))
    conf.steer_change_cost = float(rng.choice([0.0, 5.0]))
    conf.steer_angle_cost = float(rng.choice([0.0, 1.0]))
    conf.h_cost = float(rng.choice([1.0, 5.0, 15.0]))
    conf.max_steer = float(rng.choice([0.4, 0.6, 0.8]))
    conf.car.wheel_base = float(rng.choice([2.5, 3.5]))
    desc = (f"xy {conf.xy_reso} yaw {math.degrees(conf.yaw_reso):.1f} step {conf.move_step} n_steer {conf.n_steer} "
            f"check {conf.collision_check_step} ext {conf.extend_bound}")
    msz = int(rng.integers(100, 180))
    occ = W.occupancy_grid(msz, float(rng.choice([0.002, 0.005])), seed=seed + 3)
    ox, oy = W.raster_to_points(occ)
    ox, oy = np.asarray(ox, np.float64), np.asarray(oy, np.float64)
    shift = 0.0
    kind = "lattice"
    if r % 3 == 1:    # obstacle points off the lattice: the bucket-grid form of the footprint test
        ox = ox + rng.uniform(-0.3, 0.3, len(ox))
        oy = oy + rng.uniform(-0.3, 0.3, len(oy))
        kind = "points"
    elif r % 3 == 2:  # part of the map at negative coordinates (the crate's `as usize` saturates there)
        shift = -float(rng.integers(20, 60))
        ox, oy = ox + shift, oy + shift
        kind = f"lattice shifted by {shift:.0f}"
    m = api.ObstacleMap(ctx, ox, oy)
    om = O.OracleMap(ox, oy)
    desc += f", {kind}"
    # rollouts
    n = 400
    px, py = rng.uniform(3, msz - 4, n) + shift, rng.uniform(3, msz - 4, n) + shift
    pyaw = rng.uniform(-2 * math.pi, 2 * math.pi, n)
    pdir = rng.choice([-1, 1], n)
    steer, _ = api.calc_motion_set(conf)
    psteer = rng.choice(steer, n)
    pcost = rng.uniform(0, 500, n)
    got = api.expand(ctx, m, conf, px, py, pyaw, pdir, psteer, pcost)
    st, exp = om.expand(conf, px, py, pyaw, pdir, psteer, pcost)
    assert st == 0
    amb = (got["valid"] & 2) != 0
    assert amb.sum() <= 4, amb.sum()
    ok = ~amb
    assert ((got["valid"] & 1)[ok] == exp["valid"][ok]).all()
    v = ok & (exp["valid"] == 1)
    for k in ("xind", "yind", "yawind"):
        assert (got[k][v] == exp[k][v]).all(), k
    assert (got["cost"][v] == exp["cost"][v]).all()
    assert (got["end_yaw"][v] == exp["end_yaw"][v]).all()
    assert np.allclose(got["end_x"][v], exp["end_x"][v], rtol=1e-12, atol=1e-12)
    assert np.allclose(got["end_y"][v], exp["end_y"][v], rtol=1e-12, atol=1e-12)
    # analytic shots
    ns = 300
    s = np.c_[rng.uniform(3, msz - 4, ns) + shift, rng.uniform(3, msz - 4, ns) + shift, rng.uniform(-math.pi, math.pi, ns)]
    g = np.c_[rng.uniform(3, msz - 4, ns) + shift, rng.uniform(3, msz - 4, ns) + shift, rng.uniform(-math.pi, math.pi, ns)]
    g[: ns // 2] = s[: ns // 2] + rng.uniform(-25, 25, (ns // 2, 3)) * [1, 1, 0.1]
    found, idx, npaths, words, cost = api.analytic_expansion(ctx, m, conf, s, g)
    st, efound, eidx, enp, ewords, ecost = om.analytic(conf, s, g)
    oks = (found & 2) == 0
    assert ((found & 1)[oks] == (efound & 1)[oks]).all()
    assert (idx[oks] == eidx[oks]).all() and (npaths[oks] == enp[oks]).all()
    assert np.allclose(cost[oks], ecost[oks], rtol=1e-9, atol=0)
    assert (np.abs(npaths[~oks] - enp[~oks]) <= 1).all()
    # planner
    nq = 32
    starts = np.c_[rng.uniform(12, msz - 12, nq) + shift, rng.uniform(12, msz - 12, nq) + shift, rng.uniform(-3, 3, nq)]
    goals = np.c_[rng.uniform(15, msz - 15, 4) + shift, rng.uniform(15, msz - 15, 4) + shift, rng.uniform(-3, 3, 4)][rng.integers(0, 4, nq)]
    ctx.set_max_expansions(800)
    O.set_max_expansions(800)
    try:
        res = api.plan_batch(ctx, m, conf, starts, goals, cap=40000, closed_cap=1 << 12)
        differ = found_n = 0
        for i in range(nq):
            e = om.plan(conf, starts[i], goals[i])
            differ += not TP.same_batch_query(res, i, e, 1 << 12)
            found_n += res["status"][i] == 0
        errs = int((res["status"] < 0).sum())
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    desc += f", {errs} error statuses"
    say(f"config {r} ({desc}): rollouts of {n} parents equal ({int(v.sum())} valid, {int(amb.sum())} in the guard band), "
        f"{ns} shots equal ({int((~oks).sum())} flagged), planner {found_n}/{nq} found, {differ} differ after a guard-band decision")

# ---- obstacle raster + heuristic from point lists, grid A*, Reeds-Shepp trajectories
import test_gpu_astar as TA  # noqa: E402
import test_gpu_heuristic as TH  # noqa: E402
for r in range(rounds):
    seed = 0xBEE70000 + 431 * r
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100, 900))
    ex, ey = float(rng.uniform(40, 140)), float(rng.uniform(40, 140))
    lo = float(rng.choice([0.0, -30.0, 25.0]))
    ox, oy = rng.uniform(lo, lo + ex, n), rng.uniform(lo, lo + ey, n)
    if r % 2:
        ox, oy = np.round(ox), np.round(oy)  # on the lattice
    ox[:4], oy[:4] = [lo, lo + ex, lo, lo + ex], [lo, lo, lo + ey, lo + ey]
    reso = float(rng.choice([0.5, 1.0, 2.0, 2.5]))
    rr = float(rng.choice([0.3, 0.5, 1.0, 1.7])) * reso
    m = api.ObstacleMap(ctx, ox, oy)
    obs, dims = api.calc_obsmap(ctx, m, reso, rr)
    st, eobs, edims = O.obsmap(ox, oy, reso, rr, 1)
    assert dims.tolist() == edims.tolist() and (obs == eobs).all()
    gx, gy = float(rng.uniform(lo + 5, lo + ex - 5)), float(rng.uniform(lo + 5, lo + ey - 5))
    hm, _ = api.calc_holonomic_heuristic_with_obstacle(ctx, m, gx, gy, reso, rr)
    st, exp, _ = O.heuristic(ox, oy, gx, gy, reso, rr)
    assert st == 0
    TH.same_grid(hm, exp)
    paths = 0
    for _ in range(4):
        s_ = [float(rng.uniform(lo + 2, lo + ex - 2)), float(rng.uniform(lo + 2, lo + ey - 2))]
        g_ = [float(rng.uniform(lo + 2, lo + ex - 2)), float(rng.uniform(lo + 2, lo + ey - 2))]
        paths += TA.same(ctx, m, ox, oy, s_, g_, reso, rr) is not None
    # trajectories of every word of a few pairs
    maxc = float(rng.choice([0.05, 0.1, 0.1955, 0.4]))
    step = float(rng.choice([0.1, 0.2, 0.4, 1.0]))
    for _ in range(6):
        q0 = [float(rng.uniform(0, 60)), float(rng.uniform(0, 60)), float(rng.uniform(-7, 7))]
        q1 = [float(rng.uniform(0, 60)), float(rng.uniform(0, 60)), float(rng.uniform(-7, 7))]
        npth, bidx, words, flags = api.rs_batch(ctx, [q0], [q1], maxc)
        if flags[0] & 10:
            continue
        got = api.calc_all_paths(ctx, *q0, *q1, maxc, step)
        st, exp_paths = O.rs_all_paths(q0, q1, maxc, step)
        TR.check_same_paths(got, exp_paths)
    say(f"lists {r}: raster {dims.tolist()[4:]} reso {reso} rr {rr:.2f} equal, heuristic equal, grid A* {paths}/4 paths equal, "
        f"trajectories at maxc {maxc} step {step} equal")

# ---- large planner batches (two lanes, device-side tables of known cells) under drawn Configs
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
for r in range(max(1, rounds // 10)):
    seed = 0xB16B0000 + 211 * r
    rng = np.random.default_rng(seed)
    conf = A.demo_config()
    conf.xy_reso = float(rng.choice([1.5, 2.0]))
    conf.yaw_reso = math.radians(float(rng.choice([10.0, 15.0])))
    conf.n_steer = int(rng.choice([10, 20]))
    conf.h_cost = float(rng.choice([5.0, 15.0]))
    msz = 320
    occ = W.plan_map(msz, 0.02, seed=seed + 1)
    ox, oy = W.raster_to_points(occ)
    m = api.ObstacleMap(ctx, raster=occ)
    om = O.OracleMap(ox, oy)
    nq = 1200
    cand_s = np.c_[rng.uniform(10, msz - 10, 3 * nq), rng.uniform(10, msz - 10, 3 * nq), rng.uniform(-3, 3, 3 * nq)]
    free = api.collision_flags(ctx, m, conf, cand_s[:, 0], cand_s[:, 1], cand_s[:, 2]) == 0
    starts = cand_s[free][:nq]
    gp = np.c_[rng.uniform(20, msz - 20, 40), rng.uniform(20, msz - 20, 40), rng.uniform(-3, 3, 40)]
    gp = gp[api.collision_flags(ctx, m, conf, gp[:, 0], gp[:, 1], gp[:, 2]) == 0][:6]
    goals = gp[rng.integers(0, len(gp), len(starts))]
    cap = 500
    ctx.set_max_expansions(cap)
    O.set_max_expansions(cap)
    try:
        res = api.plan_batch(ctx, m, conf, starts, goals, cap=30000, closed_cap=cap)
        with ThreadPoolExecutor(8) as ex:
            exp_all = list(ex.map(lambda i: om.plan(conf, starts[i], goals[i]), range(len(starts))))
        differ = sum(not TP.same_batch_query(res, i, e, cap) for i, e in enumerate(exp_all))
    finally:
        ctx.set_max_expansions(100000)
        O.set_max_expansions(0)
    exs = np.array([int(res["stats"][i].expansions) for i in range(len(starts))])
    say(f"big batch {r}: {len(starts)} queries on a {msz}^2 map (xy {conf.xy_reso}, n_steer {conf.n_steer}): "
        f"{int((res['status'] == 0).sum())} found, {int((exs >= cap).sum())} at the cap, median {int(np.median(exs))} closed, "
        f"{differ} differ after a guard-band decision")

say("soak done")
