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
    npaths, best, words, flags = api.rs_batch(ctx, q0, q1, maxc)
    st, enp, ebest, ewords, eflags = O.rs_batch(q0, q1, maxc, threads=8)
    ok = (flags & 2) == 0
    assert ((flags & 1) == (eflags & 1)).all()
    gl, gt, gc, gn = O.words_to_numpy(words, npair)
    el, et, ec, en = O.words_to_numpy(ewords, npair)
    assert np.allclose(gt, et, rtol=1e-9, atol=0)
    assert (npaths[ok] == enp[ok]).all() and (best[ok] == ebest[ok]).all()
    assert (gc[ok] == ec[ok]).all() and (gn[ok] == en[ok]).all()
    assert np.allclose(gl[ok], el[ok], rtol=1e-9, atol=1e-9)
    bad = np.nonzero(~ok)[0]
    assert (np.abs(npaths[bad] - enp[bad]) <= 1).all()
    for i in bad[::64]:
        TR.check_flagged_pair(ctx, q0[i], q1[i], maxc, npaths[i], words[i])
    say(f"round {r}: rs_batch {npair} pairs maxc {maxc:.4f}: unflagged equal, {len(bad)} flagged within bounds")

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
    say(f"round {r}: wavefront {xw}x{yw}: bitwise equal, {int(np.isfinite(ref).sum())} cells reached")

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

say("soak done")
