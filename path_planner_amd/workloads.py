"""Synthetic inputs for the parity tests and bench.py.

All random data comes from a counter-based splitmix64 stream, value i of stream `seed`
being mix(seed + (i+1)*0x9E3779B97F4A7C15), mapped to [0,1) as (r >> 11) * 2**-53, so any
slice can be regenerated independently (per rank, on host or device) and is identical
everywhere.
"""
import math

import numpy as np

SEED_GRID = 0xC0111DE0
SEED_POSES = 0xC0111DE1
SEED_WAVE_A = 0xD1A60001
SEED_WAVE_B = 0xD1A60002
SEED_RS = 0x5EED0004
SEED_PLAN = 0x5EED0005

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, start, count):
    """uint64 values start..start+count of stream `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, start, count):
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def demo_obstacles(x=51, y=31):
    """Obstacle points of the reference demo (src/main.rs:46-84): the frame of an x-by-y
    lattice plus four wall segments; duplicates at the corners are kept as the reference
    produces them."""
    ox, oy = [], []
    ox += [float(i) for i in range(x)]; oy += [0.0] * x
    ox += [float(i) for i in range(x)]; oy += [float(y - 1)] * x
    ox += [0.0] * y; oy += [float(i) for i in range(y)]
    ox += [float(x - 1)] * y; oy += [float(i) for i in range(y)]
    ox += [float(i) for i in range(10, 21)]; oy += [15.0] * 11
    ox += [20.0] * 15; oy += [float(i) for i in range(15)]
    ox += [30.0] * 15; oy += [float(i) for i in range(15, 30)]
    ox += [40.0] * 16; oy += [float(i) for i in range(16)]
    return np.array(ox), np.array(oy)


def demo_query():
    """Start and goal of src/main.rs:233-234."""
    return (10.0, 7.0, math.radians(120.0)), (45.0, 20.0, math.radians(90.0))


def occupancy_grid(size=2048, p=0.05, seed=SEED_GRID):
    """size x size raster occ[y, x]: border occupied + interior Bernoulli(p)."""
    u = uniform01(seed, 0, size * size).reshape(size, size)
    occ = (u < p).astype(np.uint8)
    occ[0, :] = occ[-1, :] = 1
    occ[:, 0] = occ[:, -1] = 1
    return occ


def raster_to_points(occ, x0=0, y0=0):
    """Occupied cells as the reference's (ox, oy) lists, row-major order."""
    ys, xs = np.nonzero(occ)
    return xs.astype(np.float64) + x0, ys.astype(np.float64) + y0


def random_poses(n, size=2048, seed=SEED_POSES, start=0):
    """x, y ~ U[0, size-1), yaw ~ U[-pi, pi); pose i uses stream values 3i, 3i+1, 3i+2."""
    u = uniform01(seed, 3 * start, 3 * n).reshape(n, 3)
    x = u[:, 0] * (size - 1)
    y = u[:, 1] * (size - 1)
    yaw = u[:, 2] * (2.0 * math.pi) - math.pi
    return np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(yaw)


def wavefront_obsmap(n=16384, p=0.2, seed=SEED_WAVE_A, kind="bernoulli"):
    """Coarse obstacle raster obs[x, y] for the heuristic benchmark, plus the goal cell.

    kind "bernoulli": border + Bernoulli(p).  kind "walls": border + vertical walls every 64
    columns with one 3-cell gap each (long detours -> many relaxation rounds)."""
    if kind == "bernoulli":
        chunk = 1 << 24
        obs = np.empty(n * n, np.uint8)
        for s in range(0, n * n, chunk):
            c = min(chunk, n * n - s)
            obs[s:s + c] = uniform01(seed, s, c) < p
        obs = obs.reshape(n, n)
    else:
        obs = np.zeros((n, n), np.uint8)
        walls = np.arange(32, n - 1, 64)
        gaps = (splitmix64(seed, 0, len(walls)) % np.uint64(max(n - 8, 1))).astype(np.int64) + 2
        for wx, g in zip(walls, gaps):
            obs[wx, :] = 1
            obs[wx, g:g + 3] = 0
    obs[0, :] = obs[-1, :] = 1
    obs[:, 0] = obs[:, -1] = 1
    gx = gy = n // 2
    obs[gx, gy] = 0
    return obs, gx, gy


def rs_pairs(n, seed=SEED_RS, start=0, extent=100.0):
    """q0, q1 in [0,extent)^2 x [-pi,pi); pair i uses stream values 6i..6i+5."""
    u = uniform01(seed, 6 * start, 6 * n).reshape(n, 6)
    q = np.empty((n, 6))
    q[:, 0] = u[:, 0] * extent
    q[:, 1] = u[:, 1] * extent
    q[:, 2] = u[:, 2] * (2 * math.pi) - math.pi
    q[:, 3] = u[:, 3] * extent
    q[:, 4] = u[:, 4] * extent
    q[:, 5] = u[:, 5] * (2 * math.pi) - math.pi
    return q[:, :3].copy(), q[:, 3:].copy()


def demo_maxc():
    return math.tan(0.6) / 3.5


def plan_map(size=4096, density=0.006, seed=SEED_PLAN):
    """size x size raster occ[y, x] for the many-queries workload: border + sparse clustered
    obstacles (thin wall segments 8..40 cells long, random orientation) up to `density`."""
    occ = np.zeros((size, size), np.uint8)
    n_seg = int(size * size * density / 24.0)
    u = uniform01(seed, 0, 4 * n_seg).reshape(n_seg, 4)
    cx = (u[:, 0] * (size - 80)).astype(np.int64) + 40
    cy = (u[:, 1] * (size - 80)).astype(np.int64) + 40
    ln = (u[:, 2] * 33).astype(np.int64) + 8
    horiz = u[:, 3] < 0.5
    for x, y, l, h in zip(cx, cy, ln, horiz):
        if h:
            occ[y, x:x + l] = 1
        else:
            occ[y:y + l, x] = 1
    occ[0, :] = occ[-1, :] = 1
    occ[:, 0] = occ[:, -1] = 1
    return occ


def plan_queries(n, size=4096, n_goals=64, seed=SEED_PLAN + 1, min_dist=30.0, max_dist=200.0):
    """n candidate (start, goal) pose pairs; query i aims at goal i % n_goals and starts
    min_dist..max_dist away from it.  Returns (starts[n,3], goals[n,3]); the caller drops
    candidates whose start or goal pose collides."""
    ug = uniform01(seed, 0, 3 * n_goals).reshape(n_goals, 3)
    pool = np.empty((n_goals, 3))
    pool[:, 0] = 260.0 + ug[:, 0] * (size - 520.0)
    pool[:, 1] = 260.0 + ug[:, 1] * (size - 520.0)
    pool[:, 2] = ug[:, 2] * (2 * math.pi) - math.pi
    us = uniform01(seed + 1, 0, 3 * n).reshape(n, 3)
    goals = pool[np.arange(n) % n_goals]
    r = min_dist + us[:, 0] * (max_dist - min_dist)
    a = us[:, 1] * (2 * math.pi)
    starts = np.empty((n, 3))
    starts[:, 0] = goals[:, 0] + r * np.cos(a)
    starts[:, 1] = goals[:, 1] + r * np.sin(a)
    starts[:, 2] = us[:, 2] * (2 * math.pi) - math.pi
    return starts, goals.copy()
