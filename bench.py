#!/usr/bin/env python3
"""Benchmark of the hybrid A* hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload collision|rs|wavefront|queries|expand|analytic]

Default workload: BASELINE.json configs[1] -- batched footprint collision check, 16 777 216
random SE(2) poses on a 2048x2048 occupancy grid (border + Bernoulli 0.05), demo car.  One
"step" = one pass of the collision kernel over the whole pose batch, inputs resident in HBM.
Multi-GPU (--gpus N under torch.distributed.run): every rank checks its own 16M-pose slice of the
stream against a replicated grid (weak scaling, no data-path collective).

Other workloads: `rs` (configs[3], pairs sharded), `wavefront` (configs[2]; with N > 1 the grid is
cut into x-slabs, one per rank, with halo-column exchange over RCCL -- fixed total size, strong
scaling), `queries` (configs[4]: independent planner queries sharded over the ranks, strong;
`--heavy` = the search-heavy variant: denser map, longer queries), `expand` and `analytic` (the two
per-expansion kernels on their own).

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, as
fresh child processes of a parent that never touches a GPU, relays rank 0's line and exits with the
ranks' status.  When the box has fewer GPUs than ranks (a rehearsal) the ranks share the devices
and their reductions travel over gloo; the line says so under config.rehearsal.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (25 B/pose:
3 x f64 in + 1 flag byte out) / the kernel's mean launch duration measured with events on the
stream the kernel is launched on.  `cpu_baseline` = the CPU oracle (a port of the reference's
is_collision) timed on this host on a bounded sample of the same poses.

With the default workload on one GPU the same line carries, under "extra", short fixed-step legs
of the other BASELINE configs (rs 1 M pairs, wavefront 16384^2, 4096 queries on 4096^2 in its
light and its search-heavy form), each with its own ms_per_step, roofline and cpu_baseline; they
run after the timed region of the headline workload and do not touch its numbers (--no-extra
skips them).  With N > 1 the extra legs are the two workloads that shard for real -- the 4096
queries dealt by goal group (strong: the fixed list over more ranks; and weak: 4096 queries per rank)
and the 16384^2 grid in x-slabs with its halo exchange -- under a watchdog (--extra-timeout): a leg
that does not return never costs the headline line, and the run then ends with exit status 3.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak
FP64_PEAK_TFLOPS = 78.6  # MI355X vector fp64 spec peak
BYTES_PER_POSE = 25.0
BYTES_PER_PAIR = 104.0
BYTES_PER_CELL = 9.0
BYTES_PER_PRIMITIVE = 57.0 + 48.0 / 78.0  # out: 7 x 8 B + 1 flag byte; in: 48 B per parent over 78
BYTES_PER_SHOT = 121.0                    # 48 B in + 73 B out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="collision", choices=["collision", "rs", "wavefront", "queries", "expand", "analytic"])
    ap.add_argument("--poses", type=int, default=1 << 24)
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--pairs", type=int, default=1 << 20)
    ap.add_argument("--cells", type=int, default=16384, help="wavefront grid edge")
    ap.add_argument("--parents", type=int, default=1 << 18, help="expand workload: parent nodes per GPU")
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--map", type=int, default=4096, help="queries workload map edge")
    ap.add_argument("--goals", type=int, default=64, help="distinct goals the queries share")
    ap.add_argument("--max-expansions", type=int, default=0,
                    help="expansion cap per query (default 3000, with --heavy 8000: under 2 %% of its queries end there)")
    ap.add_argument("--deal", default="goal", choices=["goal", "roundrobin"],
                    help="queries workload with several ranks: a rank gets whole goal groups (it then relaxes only "
                         "its goals' heuristic grids) or every rank-th query of the list")
    ap.add_argument("--contexts", type=int, default=0,
                    help="queries workload: contexts (stream + host threads each) per GPU that share a rank's "
                         "queries through pp_plan_batch_multi, so that the latency-bound rounds of one overlap the "
                         "others' (default 1: plain pp_plan_batch, which plans a batch of this size in six parts by itself)")
    ap.add_argument("--batch-parts", type=int, default=0,
                    help="queries workload: pp_ctx_set_batch_parts (0 = the library's choice, 1 = never split)")
    ap.add_argument("--weak", action="store_true",
                    help="queries workload: --queries is the count per rank (weak scaling) instead of the total")
    ap.add_argument("--heavy", action="store_true",
                    help="queries workload: obstacle density 0.025 and starts 100-400 m from the goal "
                         "(the median query closes 74 nodes, 1.5 M nodes in all)")
    ap.add_argument("--heavy-density", type=float, default=0.025, help="obstacle density of the --heavy map")
    ap.add_argument("--band", type=float, default=0.0, help="relaxation band override (wavefront)")
    ap.add_argument("--slab-band", type=float, default=1024.0,
                    help="multi-GPU wavefront: cost band the ranks advance by per exchange round")
    ap.add_argument("--car", default="demo", choices=["demo", "compact"],
                    help="collision workload: the reference's demo car (bounding radius 5.06 cells) or a "
                         "4.0 x 1.8 m car (radius 2.46 cells: the filter's eight-row window)")
    ap.add_argument("--map-kind", default="bernoulli", choices=["bernoulli", "walls"],
                    help="wavefront obstacle raster: Bernoulli(0.2) or long walls with one gap each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs of the default run")
    ap.add_argument("--no-floors", action="store_true",
                    help="collision workload: do not run tools/k2_floors (the access-pattern floors) after the timed region")
    ap.add_argument("--extra-timeout", type=float, default=240.0,
                    help="seconds the extra legs of a multi-rank run may take before the line is printed without them")
    ap.add_argument("--cpu-sample", type=int, default=0)
    return ap.parse_args()


def measured_counters(key, units):
    """Per-launch hardware-counter figures of a workload (profiles/counters.json: HBM bytes from the
    FETCH_SIZE / WRITE_SIZE passes, fp64 operations from the SQ_INSTS_VALU_*_F64 passes), valid only
    for the size they were measured at: returned when `units` matches, else an empty dict."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            e = json.load(f).get(key) or {}
    except Exception:
        return {}
    return e if e.get("units") == units else {}


# ---------------------------------------------------------------------------------- workloads
def wl_collision(args, E):
    torch, api, W = E.torch, E.api, E.W
    occ = W.occupancy_grid(args.grid, 0.05)
    omap = api.ObstacleMap(E.ctx, raster=occ)
    n = args.poses
    x, y, yaw = E.S.pose_shard(n, args.grid, E.rank)
    dx, dy, dyaw = (torch.from_numpy(a).to(E.dev) for a in (x, y, yaw))
    flags = torch.zeros(n, dtype=torch.uint8, device=E.dev)
    wl = SimpleNamespace(
        units=n, bytes_per_unit=BYTES_PER_POSE, metric="collision_checked_poses_per_sec", unit="poses/s",
        scaling="weak", dtype="f64", kernel="k_collision", bound="hbm",
        counters_key="collision" if args.car == "demo" else None,
        config={"workload": f"footprint collision, {n} SE(2) poses/GPU on {args.grid}x{args.grid} "
                            f"occupancy grid (border + Bernoulli 0.05), {args.car} car",
                "poses_per_gpu": n, "grid": args.grid, "parallelism": f"pose-sharded x{E.world}"})
    wl.step = lambda: api.collision_flags_dev(E.ctx, omap, E.conf, dx, dy, dyaw, flags)

    def finish(out):
        # Where 8 TB/s is out of reach for this access pattern, by measurement in the same run: the
        # floor of streaming the poses alone, and of streaming them with one random L2 line per pose
        # (tools/k2_floors.hip, no arithmetic).  Uniformly random poses share no obstacle data, so
        # every per-pose lookup kernel sits on the second floor at best.
        exe = os.path.join(ROOT, "tools", "k2_floors")
        r = out["roofline"]
        r["hbm_peak_note"] = ("0.60 of the 8 TB/s figure is not reachable in this formulation: the input stream alone "
                              "runs at stream_floor_ms, and one random obstacle line per pose at gather_floor_ms")
        try:
            if E.rank != 0 or not os.path.exists(exe) or args.no_floors:
                raise FileNotFoundError(exe)
            E.torch.cuda.synchronize()
            fl = json.loads(subprocess.run([exe, str(n), "20"], capture_output=True, text=True, timeout=120,
                                           check=True).stdout.strip().splitlines()[-1])
            src = "tools/k2_floors run after the timed region"
        except Exception:
            fl = (measured_counters("collision_floors", n) or {}).get("floors")
            src = "profiles/counters.json"
        if fl:
            k = r["kernel_ms"]
            r.update({"stream_floor_ms": fl["stream_only_ms"], "gather_floor_ms": fl["stream_plus_one_line_ms"],
                      "own_pattern_floor_ms": fl["stream_plus_four_gathers_ms"],
                      "frac_of_stream_floor": fl["stream_only_ms"] / k, "frac_of_gather_floor": fl["stream_plus_one_line_ms"] / k,
                      "frac_of_own_pattern_floor": fl["stream_plus_four_gathers_ms"] / k,
                      "ceiling_frac_any_kernel": wl.units * BYTES_PER_POSE / (fl["stream_only_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "ceiling_frac_per_pose_lookup": wl.units * BYTES_PER_POSE / (fl["stream_plus_one_line_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "floors_source": src})
    wl.finish = finish

    def cpu(O):
        om = O.OracleMap(*W.raster_to_points(occ))
        ns = args.cpu_sample or min(n, 1 << 23)  # ~10 s single-threaded
        t0 = time.perf_counter()
        st, ef = om.collision_flags(E.conf, x[:ns], y[:ns], yaw[:ns])
        dt = time.perf_counter() - t0
        got = flags[:ns].cpu().numpy()
        mism = int(((got & 1) != ef).sum())
        ncpu = min(os.cpu_count() or 1, 16)
        t0 = time.perf_counter()
        om.collision_flags(E.conf, x[:ns], y[:ns], yaw[:ns], threads=ncpu)
        dtm = time.perf_counter() - t0
        # the reference's own form: KD-tree radius query, then rot2d per candidate (hybrid_astar.rs:426-445)
        nk = min(ns, 1 << 20)
        t0 = time.perf_counter()
        st, kf = om.collision_flags(E.conf, x[:nk], y[:nk], yaw[:nk], mode=2)
        dtk = time.perf_counter() - t0
        return {"value": ns / dt, "unit": wl.unit, "cores": 1, "kind": "port",
                "sample": f"first {ns} poses of the same batch, CPU oracle (is_collision restated, "
                          "bucket-grid candidate filter: faster than the reference's KD-tree form)",
                "flag_mismatches_vs_gpu": mism,
                "all_cores": {"value": ns / dtm, "cores": ncpu},
                "reference_form": {"value": nk / dtk, "cores": 1,
                                   "sample": f"first {nk} poses, KD-tree radius query + per-obstacle rot2d "
                                             "as hybrid_astar.rs:426-445 (kdtree 0.7.0's algorithm restated)",
                                   "flag_mismatches_vs_gpu": int(((got[:nk] & 1) != kf).sum())}}
    wl.cpu = cpu
    return wl


def wl_rs(args, E):
    torch, W = E.torch, E.W
    n = args.pairs
    q0, q1 = E.S.pair_shard(n, E.rank)
    cols = [torch.from_numpy(np.ascontiguousarray(q[:, k])).to(E.dev) for q in (q0, q1) for k in range(3)]
    npaths = torch.zeros(n, dtype=torch.int32, device=E.dev)
    best = torch.zeros(n, dtype=torch.int32, device=E.dev)
    words = torch.zeros(n * 56, dtype=torch.uint8, device=E.dev)
    rflags = torch.zeros(n, dtype=torch.uint8, device=E.dev)
    maxc = W.demo_maxc()
    wl = SimpleNamespace(
        units=n, bytes_per_unit=BYTES_PER_PAIR, metric="reeds_shepp_pairs_per_sec", unit="pairs/s",
        scaling="weak", dtype="f64", kernel="k_rs_batch", bound="fp64-valu", counters_key="rs",
        config={"workload": f"Reeds-Shepp all families + argmin, {n} pairs/GPU", "pairs_per_gpu": n,
                "parallelism": f"pair-sharded x{E.world}"})

    def step():
        st = E.ctx.lib.pp_rs_batch_dev(E.ctx.h, n, *[c.data_ptr() for c in cols], maxc, npaths.data_ptr(),
                                       best.data_ptr(), words.data_ptr(), rflags.data_ptr())
        assert st == 0, st
    wl.step = step

    def cpu(O):
        ns = args.cpu_sample or min(n, 1 << 18)
        t0 = time.perf_counter()
        st, enp, ebest, ewords, eflags = O.rs_batch(q0[:ns], q1[:ns], maxc, threads=1)
        dt = time.perf_counter() - t0
        fl = rflags[:ns].cpu().numpy()
        accept, tie = (fl & 2) != 0, (fl & 10) == 8
        clean = (fl & 10) == 0
        # word count: equal unless an accept / duplicate test fell inside the guard band; selected index:
        # equal unless, besides, the minimum is tied
        mism = int(((npaths[:ns].cpu().numpy() != enp) & ~accept).sum() + ((best[:ns].cpu().numpy() != ebest) & clean).sum())
        return {"value": ns / dt, "unit": wl.unit, "cores": 1, "kind": "port",
                "sample": f"first {ns} pairs, CPU oracle generate_path + argmin",
                "mismatches_vs_gpu": mism, "guard_band_pairs": int((~clean).sum()),
                "guard_band_share": {"accept_or_duplicate_test": float(accept.mean()), "tie_at_minimum_only": float(tie.mean())}}
    wl.cpu = cpu
    return wl


def wl_expand(args, E):
    torch, api, W = E.torch, E.api, E.W
    # calc_next_node for every primitive of n parents (free poses on the collision workload's grid)
    occ = W.occupancy_grid(args.grid, 0.01)
    omap = api.ObstacleMap(E.ctx, raster=occ)
    n = args.parents
    x, y, yaw = E.S.pose_shard(n, args.grid, E.rank)
    steer, _ = api.calc_motion_set(E.conf)
    P = len(steer)
    dpx, dpy, dpyaw = (torch.from_numpy(a).to(E.dev) for a in (x, y, yaw))
    dpdir = torch.ones(n, dtype=torch.int64, device=E.dev)
    dps = torch.zeros(n, dtype=torch.float64, device=E.dev)
    dpc = torch.zeros(n, dtype=torch.float64, device=E.dev)
    eout = {k: torch.zeros(n * P, dtype=(torch.uint8 if k == "valid" else torch.int64 if k in
                                         ("xind", "yind", "yawind") else torch.float64), device=E.dev)
            for k in ("valid", "xind", "yind", "yawind", "cost", "end_x", "end_y", "end_yaw")}
    wl = SimpleNamespace(
        units=n * P, bytes_per_unit=BYTES_PER_PRIMITIVE, metric="motion_primitive_rollouts_per_sec",
        unit="primitives/s", scaling="weak", dtype="f64", kernel="k_expand", bound="fp64-valu",
        counters_key="expand",
        config={"workload": f"calc_next_node: {n} parents/GPU x {P} primitives (11-pose rollout, 3 poses "
                            f"collision-checked) on a {args.grid}x{args.grid} grid (Bernoulli 0.01)",
                "parents_per_gpu": n, "primitives": P, "parallelism": f"parent-sharded x{E.world}"})
    wl.step = lambda: api.expand_dev(E.ctx, omap, E.conf, dpx, dpy, dpyaw, dpdir, dps, dpc, eout)

    def cpu(O):
        ns = args.cpu_sample or min(n, 1 << 15)
        om = O.OracleMap(*W.raster_to_points(occ))
        t0 = time.perf_counter()
        st, eo = om.expand(E.conf, x[:ns], y[:ns], yaw[:ns], np.ones(ns, np.int64), np.zeros(ns), np.zeros(ns))
        dt = time.perf_counter() - t0
        gv = eout["valid"][: ns * P].cpu().numpy()
        mism = int((((gv & 1) != (eo["valid"] & 1)) & ((gv & 2) == 0)).sum())
        return {"value": ns * P / dt, "unit": wl.unit, "cores": 1, "kind": "port",
                "sample": f"first {ns} parents, CPU oracle calc_next_node", "valid_mismatches_vs_gpu": mism}
    wl.cpu = cpu
    return wl


def wl_analytic(args, E):
    torch, api, W = E.torch, E.api, E.W
    # analystic_expansion for n (node, goal) pairs: all words, cost order, sampled poses of each
    # path collision-checked until one is free.  Device-pointer entry point: 48 B in / 73 B out
    # per pair stay in HBM.
    occ = W.occupancy_grid(args.grid, 0.01)
    omap = api.ObstacleMap(E.ctx, raster=occ)
    n = args.pairs // 4
    q0, q1 = E.S.pair_shard(n, E.rank, extent=200.0)
    q0[:, :2] += args.grid / 2 - 100.0
    q1[:, :2] += args.grid / 2 - 100.0
    q6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q0.T, q1.T]))).to(E.dev)
    a_found = torch.zeros(n, dtype=torch.uint8, device=E.dev)
    a_idx = torch.zeros(n, dtype=torch.int32, device=E.dev)
    a_np = torch.zeros(n, dtype=torch.int32, device=E.dev)
    a_words = torch.zeros((n, 56), dtype=torch.uint8, device=E.dev)
    a_cost = torch.zeros(n, dtype=torch.float64, device=E.dev)
    wl = SimpleNamespace(
        units=n, bytes_per_unit=BYTES_PER_SHOT, metric="analytic_expansion_shots_per_sec", unit="shots/s",
        scaling="weak", dtype="f64", kernel="k_rs_words+k_rs_shot", bound="fp64-valu", counters_key="analytic",
        config={"workload": f"analytic expansion: {n} (node, goal) pairs/GPU within a 200 m box on a "
                            f"{args.grid}x{args.grid} grid (Bernoulli 0.01), demo car and costs",
                "pairs_per_gpu": n, "parallelism": f"pair-sharded x{E.world}"})
    wl.step = lambda: api.analytic_expansion_dev(E.ctx, omap, E.conf, q6, a_found, a_idx, a_np, a_words, a_cost)

    def cpu(O):
        ns = args.cpu_sample or min(n, 4096)
        om = O.OracleMap(*W.raster_to_points(occ))
        t0 = time.perf_counter()
        st, efound, eidx, enp, ewords, ecost = om.analytic(E.conf, q0[:ns], q1[:ns])
        dt = time.perf_counter() - t0
        gfound, gidx = a_found[:ns].cpu().numpy(), a_idx[:ns].cpu().numpy()
        clean = (gfound & 2) == 0
        mism = int((((gfound & 1) != (efound & 1)) & clean).sum() +
                   (((gfound & 1) == 1) & ((efound & 1) == 1) & (gidx != eidx) & clean).sum())
        return {"value": ns / dt, "unit": wl.unit, "cores": 1, "kind": "port",
                "sample": f"first {ns} pairs, CPU oracle analystic_expansion", "mismatches_vs_gpu": mism,
                "found_fraction": float((efound & 1).mean())}
    wl.cpu = cpu
    return wl


def wl_wavefront(args, E):
    torch, api, W = E.torch, E.api, E.W
    from path_planner_amd import tiled
    ncell = args.cells
    obs_np, gx, gy = W.wavefront_obsmap(ncell, 0.2, kind=args.map_kind)
    x0, x1 = tiled.slab_bounds(ncell, E.world, E.rank)
    mk = "Bernoulli 0.2" if args.map_kind == "bernoulli" else "a wall every 64 columns with one 3-cell gap"
    wl = SimpleNamespace(
        units=ncell * ncell // E.world, bytes_per_unit=BYTES_PER_CELL, metric="heuristic_cells_per_sec",
        unit="cells/s", scaling="strong", dtype="f64", kernel="k_tile_round+k_tile_list", bound="hbm",
        counters_key=f"wavefront{ncell}" if args.map_kind == "bernoulli" else f"wavefront{ncell}_{args.map_kind}",
        config={"workload": f"holonomic heuristic wavefront, {ncell}x{ncell} grid (border + {mk}), goal at centre",
                "cells": ncell * ncell,
                "parallelism": f"x-slabs x{E.world}, halo-column exchange" if E.world > 1 else "1 GPU"})
    wl.bound_note = ("priced against HBM as the 9 B/cell minimum asks; what the time is made of is ~430 cost-ordered "
                     "rounds of one capped tile visit each (load 5 us, sweeps 12 us, store 3 us) plus a 5 us list kernel, "
                     "the crowded rounds in the middle VALU-bound on the fp64 sweeps")
    if E.world == 1:
        obs = torch.from_numpy(obs_np).to(E.dev)
        cost = torch.empty((ncell, ncell), dtype=torch.float64, device=E.dev)
        wl.step = lambda: api.wavefront_dev(E.ctx, obs, gx, gy, cost)
        rounds = None
    else:
        obs_s, cost = tiled.make_slab(obs_np, ncell, ncell, x0, x1, E.dev)
        rounds = []
        solver = tiled.DeviceSlabSolver(E.ctx, ncell, ncell, x0, x1, obs_s, (gx, gy))

        def step():
            # packing the raster and writing the dense slab are part of the job, as they are on one GPU
            solver.reset(obs_s)
            rounds.append(tiled.tiled_wavefront(solver, E.rank, E.world, band=args.slab_band))
            solver.read(cost)
        wl.step = step

    def finish(out):
        out["config"]["relaxation"] = api.wavefront_stats(E.ctx)
        if rounds:
            out["config"]["exchange_rounds"] = rounds[-1]
    wl.finish = finish

    def cpu(O):
        ns = args.cpu_sample or min(ncell, 2048)
        sub = np.ascontiguousarray(obs_np[:ns, :ns]).copy()
        sub[0, :] = sub[-1, :] = 1
        sub[:, 0] = sub[:, -1] = 1
        sub[ns // 2, ns // 2] = 0
        t0 = time.perf_counter()
        st, ref = O.wavefront(sub, ns // 2, ns // 2)
        dt = time.perf_counter() - t0
        res = {"value": ns * ns / dt, "unit": wl.unit, "cores": 1, "kind": "port",
               "sample": f"{ns}x{ns} corner of the same raster, CPU oracle Dijkstra (dense arrays, binary heap)"}
        if E.world == 1:
            # the same corner through the device path: bitwise equal
            d_sub = torch.from_numpy(sub).to(E.dev)
            d_cost = torch.empty((ns, ns), dtype=torch.float64, device=E.dev)
            api.wavefront_dev(E.ctx, d_sub, ns // 2, ns // 2, d_cost)
            E.ctx.synchronize()
            got = d_cost.cpu().numpy()
            res["cell_mismatches_vs_gpu"] = int((got.view(np.uint64) != ref.view(np.uint64)).sum())
        return res
    wl.cpu = cpu
    return wl


def wl_queries(args, E):
    api, W, A = E.api, E.W, E.A
    dens, dmin, dmax = (args.heavy_density, 100.0, 400.0) if args.heavy else (0.006, 30.0, 200.0)
    max_exp = args.max_expansions or (8000 if args.heavy else 3000)
    occ = W.plan_map(args.map, dens)
    omap = api.ObstacleMap(E.ctx, raster=occ)
    E.ctx.set_max_expansions(max_exp)
    E.ctx.set_batch_parts(args.batch_parts or int(os.environ.get("PP_BENCH_BATCH_PARTS", "0")))
    # candidates whose start or goal pose collides are dropped (device check), then the
    # first --queries survivors are dealt round-robin to the ranks
    total = args.queries * (E.world if args.weak else 1)
    cs, cg = W.plan_queries((8 if args.heavy else 3) * total, args.map, args.goals, min_dist=dmin, max_dist=dmax)
    inside = (cs[:, 0] > 8) & (cs[:, 0] < args.map - 9) & (cs[:, 1] > 8) & (cs[:, 1] < args.map - 9)
    fs = api.collision_flags(E.ctx, omap, E.conf, cs[:, 0], cs[:, 1], cs[:, 2])
    fg = api.collision_flags(E.ctx, omap, E.conf, cg[:, 0], cg[:, 1], cg[:, 2])
    keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:total]
    assert len(keep) == total, "not enough collision-free candidates"
    if args.deal == "goal":
        # the planner relaxes one heuristic grid per distinct goal cell (astar.rs:141-146) of its batch
        cells = np.round(cg[keep][:, :2] / E.conf.xy_reso).astype(np.int64)
        mine = keep[E.S.query_shard_by_goal(cells, E.rank, E.world)]
    else:
        mine = E.S.query_shard(keep, E.rank, E.world)
    starts, goals = cs[mine], cg[mine]
    wl = SimpleNamespace(
        units=len(mine), bytes_per_unit=0.0, metric="planning_queries_per_sec", unit="queries/s",
        scaling="weak" if args.weak else "strong", dtype="f64", kernel="k_round / k_rs_words+k_rs_shot+k_expand",
        bound="latency", counters_key=None,
        config={"workload": f"{total} hybrid A* queries on a {args.map}x{args.map} map (border + "
                            f"clustered wall segments, density {occ.mean():.4f}), {args.goals} shared goals, "
                            f"start {dmin:.0f}-{dmax:.0f} m from goal, expansion cap {max_exp}",
                "queries": total, "variant": "search-heavy" if args.heavy else "light",
                "parallelism": f"query-sharded x{E.world}",
                "deal": ("whole goal groups per rank" if args.deal == "goal" else "round-robin") if E.world > 1 else None})
    last = {}
    K = args.contexts if args.contexts > 0 else 1
    wl.config["contexts_per_gpu"] = K
    if K > 1:
        base_threads = E.host_threads
        ctxs = [E.ctx] + [api.Context(E.ctx.device) for _ in range(K - 1)]
        maps = [omap] + [api.ObstacleMap(c, raster=occ) for c in ctxs[1:]]
        for c in ctxs:
            c.set_max_expansions(max_exp)
            c.set_host_threads(max(1, base_threads // K))

    def step():
        if K > 1:
            last["r"] = api.plan_batch(ctxs, maps, E.conf, starts, goals, cap=4096, out=last.get("r"))
        else:
            last["r"] = api.plan_batch(E.ctx, omap, E.conf, starts, goals, cap=4096, out=last.get("r"))
    wl.step = step

    def finish(out):
        # host-driven search: no single kernel bounds it; report the search's own counters
        r = last["r"]
        stt = r["stats"]
        nq = len(r["status"])
        ex = np.array([int(stt[i].expansions) for i in range(nq)])
        out["search"] = {
            "found": int((r["status"] == 0).sum()), "no_path": int((r["status"] == 1).sum()),
            "cap_reached": int((r["status"] == A.PP_ERR_LIMIT).sum()),
            "other_errors": int(((r["status"] < 0) & (r["status"] != A.PP_ERR_LIMIT)).sum()),
            "expansions": int(ex.sum()), "expansions_median": float(np.median(ex)),
            "expansions_per_sec": float(ex.sum() / (out["ms_per_step"] * 1e-3)),
            "poses_checked": int(sum(stt[i].poses_checked for i in range(nq))),
            "rs_paths": int(sum(stt[i].rs_paths for i in range(nq))),
            "t_heuristic_ms": float(stt[0].t_heuristic_ms), "t_search_ms": float(stt[0].t_search_ms),
            "note": "counters of rank 0's shard, last step"}
        # what sets the time of a lock-step batch, per rank: the grids it relaxes, its rounds (and how many of
        # them ran with the chip nearly empty) and its longest query
        cs_ = [api.plan_batch_counters(c) for c in (ctxs if K > 1 else [E.ctx])]
        agg = {k: (sum if k in ("nodes_sent", "goal_grids", "expansions", "queries", "lanes") else max)(c[k] for c in cs_)
               for k in cs_[0]}
        # (t_heuristic_ms / t_search_ms are stamped per context: the largest of them is the rank's)
        mine_c = dict(agg, contexts=len(cs_), t_heuristic_ms=float(max(s_.t_heuristic_ms for s_ in stt)),
                      t_search_ms=float(max(s_.t_search_ms for s_ in stt)),
                      cap_reached=int((r["status"] == A.PP_ERR_LIMIT).sum()))
        E.ctx.set_host_threads(E.host_threads)
        if E.world > 1:
            allc = [None] * E.world
            E.dist.all_gather_object(allc, mine_c)
            out["search"]["per_rank"] = allc
        else:
            out["search"]["per_rank"] = [mine_c]
    wl.finish = finish

    def cpu(O):
        from concurrent.futures import ThreadPoolExecutor
        nt = args.cpu_sample or (16 if args.heavy else 32)  # timed one after the other on one core (10-20 s)
        nc = 64                        # checked against the device result (the rest on a thread pool)
        om = O.OracleMap(*W.raster_to_points(occ))
        O.set_max_expansions(max_exp)
        r = last["r"]

        def differs(i, e):
            ok = e["status"] == r["status"][i] and e["stats"].expansions == r["stats"][i].expansions
            if ok and e["status"] == 0:
                ok = abs(e["cost"] - r["cost"][i]) <= 1e-9 * abs(e["cost"]) and len(e["x"]) == r["len"][i]
            return (not ok) and r["stats"][i].ambiguous == 0
        t0 = time.perf_counter()
        first = [om.plan(E.conf, starts[i], goals[i]) for i in range(nt)]
        dt = time.perf_counter() - t0
        mism = sum(differs(i, e) for i, e in enumerate(first))
        with ThreadPoolExecutor(min(os.cpu_count() or 1, 16)) as ex:
            rest = list(ex.map(lambda i: om.plan(E.conf, starts[i], goals[i]), range(nt, nc)))
        mism += sum(differs(nt + k, e) for k, e in enumerate(rest))
        O.set_max_expansions(0)
        return {"value": nt / dt, "unit": wl.unit, "cores": 1, "kind": "port",
                "sample": f"first {nt} queries of rank 0's shard timed on one core, CPU oracle planner (heuristic "
                          f"recomputed per query as the reference does); {nc} queries compared with the device result",
                "queries_compared": nc, "mismatches_vs_gpu": int(mism)}
    wl.cpu = cpu
    return wl


BUILDERS = {"collision": wl_collision, "rs": wl_rs, "expand": wl_expand, "analytic": wl_analytic,
            "wavefront": wl_wavefront, "queries": wl_queries}


def roofline(wl, dev_ms):
    """The bound that applies to the workload's dominant kernel, named for what it is."""
    hbm = wl.units * wl.bytes_per_unit / (dev_ms * 1e-3) / 1e9 if wl.bytes_per_unit else None
    c = measured_counters(wl.counters_key, wl.units) if wl.counters_key else {}
    if wl.bound == "hbm":
        r = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
             "traffic": c.get("hbm_bytes"), "kernel": wl.kernel, "kernel_ms": dev_ms,
             "bytes_per_unit": wl.bytes_per_unit}
        if getattr(wl, "bound_note", None):
            r["note"] = wl.bound_note
        return r
    if wl.bound == "fp64-valu":
        fl = c.get("fp64_flops")
        tf = fl / (dev_ms * 1e-3) / 1e12 if fl else None
        return {"bound": "fp64-valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / FP64_PEAK_TFLOPS if tf else None, "traffic": c.get("hbm_bytes"),
                "kernel": wl.kernel, "kernel_ms": dev_ms,
                "flops_per_launch": fl, "flops_source": c.get("source"),
                "hbm_gbs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS if hbm else None, "bytes_per_unit": wl.bytes_per_unit}
    return {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
            "kernel": wl.kernel, "kernel_ms": dev_ms,
            "note": "host-driven lock-step search: a round is one small launch plus host heap work; "
                    "neither HBM nor the ALUs bound it (see the search counters)"}


def self_launch(args):
    """`--gpus N` with no launcher around it: start the N ranks as fresh processes (torch's own
    launcher, one rank per GPU), relay what rank 0 prints and leave with the ranks' exit status.
    This process never initialises a GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = 0
    for ln in child.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        lines += ln.startswith("{")
    rc = child.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks printed {lines} result lines, expected 1", file=sys.stderr)
        rc = 4
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    from path_planner_amd import _abi as A
    from path_planner_amd import api
    from path_planner_amd import sharding as S
    from path_planner_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # A box with fewer GPUs than ranks can only rehearse the multi-rank path: the ranks share the
    # devices round-robin and the reductions travel through host memory (gloo).  PP_BENCH_BACKEND
    # forces either backend.
    ndev = max(torch.cuda.device_count(), 1)  # counting devices does not initialise one
    local_world_env = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    backend = os.environ.get("PP_BENCH_BACKEND") or ("nccl" if ndev >= local_world_env else "gloo")
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    rdev = dev if backend == "nccl" else torch.device("cpu")  # where reduction tensors live
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    stream = torch.cuda.Stream(device=dev)
    ctx = api.Context(dev_index, stream.cuda_stream)
    # the planner's worker threads spin between rounds: several ranks on one node share its cores
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    host_threads = 15
    if local_world > 1:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 16)
        host_threads = max(1, min(15, cores // local_world - 1))
        ctx.set_host_threads(host_threads)
    conf = A.demo_config()
    if args.car == "compact":
        conf.car.center2front, conf.car.center2back, conf.car.width, conf.extend_bound = 3.0, 1.0, 1.8, 0.2
    if args.band > 0:
        assert ctx.lib.pp_ctx_set_relax_band(ctx.h, args.band) == 0
    E = SimpleNamespace(torch=torch, api=api, W=W, A=A, S=S, ctx=ctx, conf=conf, dev=dev, rank=rank, world=world,
                        dist=dist, host_threads=host_threads)
    rehearsal = (f"{world} ranks on {ndev} GPU(s), reductions over {backend}: a rehearsal of the multi-rank path, "
                 "not a scaling measurement") if world > 1 and (backend != "nccl" or ndev < local_world_env) else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(wl, steps, warmup):
        """W untimed steps, then exactly K steps between barriers; (wall seconds, device ms per step)."""
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                wl.step()
            barrier()
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(steps):
                wl.step()
            ev1.record(stream)
            barrier()
            t1 = time.perf_counter()
        return t1 - t0, ev0.elapsed_time(ev1) / steps

    def line(wl, a, steps, warmup, elapsed, dev_ms, total_units):
        out = {"metric": wl.metric, "value": total_units / elapsed, "unit": wl.unit, "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
               "scaling": wl.scaling, "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
               "config": wl.config, "roofline": roofline(wl, dev_ms)}
        if rehearsal:
            out["config"]["rehearsal"] = rehearsal
        if getattr(wl, "finish", None):
            wl.finish(out)
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            out["cpu_baseline"] = wl.cpu(O)
        return out

    # ------------------------------------------------------------------ the headline workload
    wl = BUILDERS[args.workload](args, E)
    elapsed, dev_ms = timed(wl, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tu = torch.tensor([wl.units], dtype=torch.int64, device=rdev)
        dist.all_reduce(tu)
        total_units = int(tu.item()) * args.steps
    else:
        total_units = wl.units * args.steps
    out = line(wl, args, args.steps, args.warmup, elapsed, dev_ms, total_units)

    # ------------------------------------------------------------------ the other configs, briefly
    if args.workload == "collision" and world == 1 and not args.no_extra:
        del wl
        torch.cuda.empty_cache()
        out["extra"] = {}
        legs = [("rs", {"workload": "rs"}, 10, 2), ("wavefront", {"workload": "wavefront"}, 2, 1),
                ("queries", {"workload": "queries"}, 6, 2), ("queries_heavy", {"workload": "queries", "heavy": True}, 2, 1)]
        for name, over, steps, warmup in legs:
            a = argparse.Namespace(**{**vars(parse_defaults()), **over, "no_cpu_baseline": args.no_cpu_baseline})
            try:
                w2 = BUILDERS[a.workload](a, E)
                el, dms = timed(w2, steps, warmup)
                out["extra"][name] = line(w2, a, steps, warmup, el, dms, w2.units * steps)
                del w2
            except Exception as ex:  # a leg that fails must not take the headline number with it
                out["extra"][name] = {"error": f"{type(ex).__name__}: {ex}"}
            ctx.set_max_expansions(100000)
            torch.cuda.empty_cache()
    # With several ranks the two workloads that are sharded for real follow: the 4096 queries dealt
    # round-robin and the 16384^2 heuristic grid in x-slabs with its halo exchange.  They run under a
    # watchdog: if a leg has not returned in time (a collective that never completes), every rank
    # leaves by itself and rank 0 prints the line with what it has, the headline number first of all.
    printed = threading.Lock()

    def emit():
        if printed.acquire(blocking=False) and rank == 0:
            print(json.dumps(out), flush=True)

    if args.workload == "collision" and world > 1 and not args.no_extra:
        del wl
        torch.cuda.empty_cache()
        out["extra"] = {}

        in_flight = ["(none)"]

        def give_up():
            # a process that has used the GPU and gives up on a collective must not report success:
            # the line (headline first) still goes out, then every rank leaves with status 3.  Rank 0
            # goes first so that the launcher, which ends the others when one rank fails, cannot cut
            # its line off.
            out["extra"]["watchdog"] = f"extra legs stopped after {args.extra_timeout} s, in leg '{in_flight[0]}'"
            emit()
            if rank != 0:
                time.sleep(5.0)
            os._exit(3)
        dog = threading.Timer(args.extra_timeout, give_up)
        dog.daemon = True
        dog.start()
        for name, over, steps, warmup in [("queries", {"workload": "queries"}, 6, 2),
                                          ("queries_weak", {"workload": "queries", "weak": True}, 6, 2),
                                          ("wavefront", {"workload": "wavefront"}, 2, 1)]:
            in_flight[0] = name
            # sizes as given on the command line (the BASELINE sizes unless a rehearsal shrinks them)
            a = argparse.Namespace(**{**vars(args), **over, "no_cpu_baseline": True})
            try:
                w2 = BUILDERS[a.workload](a, E)
                el, dms = timed(w2, steps, warmup)
                t = torch.tensor([el], dtype=torch.float64, device=rdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tu = torch.tensor([w2.units], dtype=torch.int64, device=rdev)
                dist.all_reduce(tu)
                out["extra"][name] = line(w2, a, steps, warmup, float(t.item()), dms, int(tu.item()) * steps)
                del w2
            except Exception as ex:  # the same failure on every rank: the next leg still runs
                out["extra"][name] = {"error": f"{type(ex).__name__}: {ex}"}
            ctx.set_max_expansions(100000)
            torch.cuda.empty_cache()
        dog.cancel()
    emit()
    if world > 1:
        dist.destroy_process_group()


def parse_defaults():
    saved = sys.argv
    sys.argv = [saved[0]]
    try:
        return parse()
    finally:
        sys.argv = saved


if __name__ == "__main__":
    main()
