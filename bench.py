#!/usr/bin/env python3
"""Benchmark of the hybrid A* hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload collision|rs|wavefront|queries]

Default workload: BASELINE.json configs[1] -- batched footprint collision check, 16 777 216
random SE(2) poses on a 2048x2048 occupancy grid (border + Bernoulli 0.05), demo car.  One
"step" = one pass of the collision kernel over the whole pose batch, inputs resident in HBM.
Multi-GPU (--gpus N under torch.distributed.run): every rank checks its own 16M-pose slice of the
stream against a replicated grid (weak scaling, no data-path collective).

Other workloads: `rs` (configs[3], pairs sharded), `wavefront` (configs[2]; with N > 1 the grid is
cut into x-slabs, one per rank, with halo-column exchange over RCCL -- fixed total size, strong
scaling) and `queries` (configs[4]: independent planner queries sharded over the ranks, strong).

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (25 B/pose:
3 x f64 in + 1 flag byte out) / the kernel's mean launch duration measured with events on the
stream the kernel is launched on.  `cpu_baseline` = the CPU oracle (a port of the reference's
is_collision) timed on this host on a bounded sample of the same poses.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
BYTES_PER_POSE = 25.0
BYTES_PER_PAIR = 104.0
BYTES_PER_CELL = 9.0
BYTES_PER_PRIMITIVE = 57.0 + 48.0 / 78.0  # out: 7 x 8 B + 1 flag byte; in: 48 B per parent over 78


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
    ap.add_argument("--max-expansions", type=int, default=3000)
    ap.add_argument("--band", type=float, default=0.0, help="relaxation band override (wavefront)")
    ap.add_argument("--map-kind", default="bernoulli", choices=["bernoulli", "walls"],
                    help="wavefront obstacle raster: Bernoulli(0.2) or long walls with one gap each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    return ap.parse_args()


def traffic_from_profile(name):
    """HBM bytes per launch measured with rocprofv3 PMC passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(name)
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from path_planner_amd import _abi as A
    from path_planner_amd import api
    from path_planner_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks:
    # ranks share the devices round-robin and the reductions travel through host memory.
    backend = os.environ.get("PP_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
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
    conf = A.demo_config()
    if args.band > 0:
        assert ctx.lib.pp_ctx_set_relax_band(ctx.h, args.band) == 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ set-up
    scaling = "weak"
    if args.workload == "collision":
        occ = W.occupancy_grid(args.grid, 0.05)
        omap = api.ObstacleMap(ctx, raster=occ)
        n = args.poses
        x, y, yaw = W.random_poses(n, args.grid, start=rank * n)
        dx, dy, dyaw = (torch.from_numpy(a).to(dev) for a in (x, y, yaw))
        flags = torch.zeros(n, dtype=torch.uint8, device=dev)
        units, bytes_per_unit = n, BYTES_PER_POSE
        metric, unit = "collision_checked_poses_per_sec", "poses/s"
        wl = {"workload": f"footprint collision, {n} SE(2) poses/GPU on {args.grid}x{args.grid} "
                          "occupancy grid (border + Bernoulli 0.05), demo car",
              "poses_per_gpu": n, "grid": args.grid, "parallelism": f"pose-sharded x{world}"}
        dtype, kernel = "f64", "k_collision"

        def step():
            api.collision_flags_dev(ctx, omap, conf, dx, dy, dyaw, flags)
    elif args.workload == "rs":
        n = args.pairs
        q0, q1 = W.rs_pairs(n, start=rank * n)
        cols = [torch.from_numpy(np.ascontiguousarray(q[:, k])).to(dev) for q in (q0, q1) for k in range(3)]
        npaths = torch.zeros(n, dtype=torch.int32, device=dev)
        best = torch.zeros(n, dtype=torch.int32, device=dev)
        words = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
        rflags = torch.zeros(n, dtype=torch.uint8, device=dev)
        maxc = W.demo_maxc()
        units, bytes_per_unit = n, BYTES_PER_PAIR
        metric, unit = "reeds_shepp_pairs_per_sec", "pairs/s"
        wl = {"workload": f"Reeds-Shepp all families + argmin, {n} pairs/GPU", "pairs_per_gpu": n,
              "parallelism": f"pair-sharded x{world}"}
        dtype, kernel = "f64", "k_rs_batch"

        def step():
            st = ctx.lib.pp_rs_batch_dev(ctx.h, n, *[c.data_ptr() for c in cols], maxc,
                                         npaths.data_ptr(), best.data_ptr(), words.data_ptr(),
                                         rflags.data_ptr())
            assert st == 0, st
    elif args.workload == "expand":
        # calc_next_node for every primitive of n parents (free poses on the collision workload's grid)
        occ = W.occupancy_grid(args.grid, 0.01)
        omap = api.ObstacleMap(ctx, raster=occ)
        n = args.parents
        x, y, yaw = W.random_poses(n, args.grid, start=rank * n)
        steer, _ = api.calc_motion_set(conf)
        P = len(steer)
        dpx, dpy, dpyaw = (torch.from_numpy(a).to(dev) for a in (x, y, yaw))
        dpdir = torch.ones(n, dtype=torch.int64, device=dev)
        dps = torch.zeros(n, dtype=torch.float64, device=dev)
        dpc = torch.zeros(n, dtype=torch.float64, device=dev)
        eout = {k: torch.zeros(n * P, dtype=(torch.uint8 if k == "valid" else torch.int64 if k in
                                             ("xind", "yind", "yawind") else torch.float64), device=dev)
                for k in ("valid", "xind", "yind", "yawind", "cost", "end_x", "end_y", "end_yaw")}
        units, bytes_per_unit = n * P, BYTES_PER_PRIMITIVE
        metric, unit = "motion_primitive_rollouts_per_sec", "primitives/s"
        wl = {"workload": f"calc_next_node: {n} parents/GPU x {P} primitives (11-pose rollout, 3 poses "
                          f"collision-checked) on a {args.grid}x{args.grid} grid (Bernoulli 0.01)",
              "parents_per_gpu": n, "primitives": P, "parallelism": f"parent-sharded x{world}"}
        dtype, kernel = "f64", "k_expand"

        def step():
            api.expand_dev(ctx, omap, conf, dpx, dpy, dpyaw, dpdir, dps, dpc, eout)
    elif args.workload == "analytic":
        # analystic_expansion for n (node, goal) pairs: all words, cost order, sampled poses of each
        # path collision-checked until one is free.  Device-pointer entry point: 48 B in / 73 B out
        # per pair stay in HBM.
        occ = W.occupancy_grid(args.grid, 0.01)
        omap = api.ObstacleMap(ctx, raster=occ)
        n = args.pairs // 4
        q0, q1 = W.rs_pairs(n, start=rank * n, extent=200.0)
        q0[:, :2] += args.grid / 2 - 100.0
        q1[:, :2] += args.grid / 2 - 100.0
        units, bytes_per_unit = n, 121.0
        metric, unit = "analytic_expansion_shots_per_sec", "shots/s"
        wl = {"workload": f"analytic expansion: {n} (node, goal) pairs/GPU within a 200 m box on a "
                          f"{args.grid}x{args.grid} grid (Bernoulli 0.01), demo car and costs",
              "pairs_per_gpu": n, "parallelism": f"pair-sharded x{world}"}
        dtype, kernel = "f64", "k_rs_words+k_rs_shot"
        q6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q0.T, q1.T]))).to(dev)
        a_found = torch.zeros(n, dtype=torch.uint8, device=dev)
        a_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        a_np = torch.zeros(n, dtype=torch.int32, device=dev)
        a_words = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
        a_cost = torch.zeros(n, dtype=torch.float64, device=dev)
        last = {}

        def step():
            api.analytic_expansion_dev(ctx, omap, conf, q6, a_found, a_idx, a_np, a_words, a_cost)
    elif args.workload == "wavefront":
        from path_planner_amd import tiled
        ncell = args.cells
        obs_np, gx, gy = W.wavefront_obsmap(ncell, 0.2, kind=args.map_kind)
        x0, x1 = tiled.slab_bounds(ncell, world, rank)
        units, bytes_per_unit = ncell * ncell // world, BYTES_PER_CELL
        metric, unit = "heuristic_cells_per_sec", "cells/s"
        scaling = "strong"
        mk = "Bernoulli 0.2" if args.map_kind == "bernoulli" else "a wall every 64 columns with one 3-cell gap"
        wl = {"workload": f"holonomic heuristic wavefront, {ncell}x{ncell} grid (border + "
                          f"{mk}), goal at centre", "cells": ncell * ncell,
              "parallelism": f"x-slabs x{world}, halo-column exchange" if world > 1 else "1 GPU"}
        dtype, kernel = "f64", "k_tile_round+k_tile_list"
        if world == 1:
            obs = torch.from_numpy(obs_np).to(dev)
            cost = torch.empty((ncell, ncell), dtype=torch.float64, device=dev)

            def step():
                api.wavefront_dev(ctx, obs, gx, gy, cost)
        else:
            obs_s, cost = tiled.make_slab(obs_np, ncell, ncell, x0, x1, dev)
            solver = tiled.DeviceSlabSolver(ctx, ncell, ncell, x0, x1, obs_s, cost, (gx, gy))
            rounds = []

            def step():
                cost.fill_(float("inf"))
                torch.cuda.synchronize()
                rounds.append(tiled.tiled_wavefront(solver, rank, world))
    else:
        occ = W.plan_map(args.map)
        omap = api.ObstacleMap(ctx, raster=occ)
        ctx.set_max_expansions(args.max_expansions)
        # candidates whose start or goal pose collides are dropped (device check), then the
        # first --queries survivors are dealt round-robin to the ranks
        cs, cg = W.plan_queries(3 * args.queries, args.map, args.goals)
        inside = (cs[:, 0] > 8) & (cs[:, 0] < args.map - 9) & (cs[:, 1] > 8) & (cs[:, 1] < args.map - 9)
        fs = api.collision_flags(ctx, omap, conf, cs[:, 0], cs[:, 1], cs[:, 2])
        fg = api.collision_flags(ctx, omap, conf, cg[:, 0], cg[:, 1], cg[:, 2])
        keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][: args.queries]
        assert len(keep) == args.queries, "not enough collision-free candidates"
        mine = keep[rank::world]
        starts, goals = cs[mine], cg[mine]
        units, bytes_per_unit = len(mine), 0.0
        metric, unit = "planning_queries_per_sec", "queries/s"
        scaling = "strong"
        wl = {"workload": f"{args.queries} hybrid A* queries on a {args.map}x{args.map} map (border + "
                          f"clustered wall segments, density {occ.mean():.4f}), {args.goals} shared goals, "
                          f"start 30-200 m from goal, expansion cap {args.max_expansions}",
              "queries": args.queries, "parallelism": f"query-sharded x{world}"}
        dtype, kernel = "f64", "k_expand+k_rs_shot"
        last = {}

        def step():
            last["r"] = api.plan_batch(ctx, omap, conf, starts, goals, cap=4096, out=last.get("r"))

    # ------------------------------------------------------------------ timed region
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps  # mean duration of one launch on its stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if world > 1:
        tu = torch.tensor([units], dtype=torch.int64, device=rdev)
        dist.all_reduce(tu)
        total_units = int(tu.item()) * args.steps
    else:
        total_units = units * args.steps
    value = total_units / elapsed

    out = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": wl,
    }
    achieved = units * bytes_per_unit / (dev_ms * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(f"wavefront{args.cells}" if args.workload == "wavefront" and args.map_kind == "bernoulli" else args.workload),
                       "kernel": kernel, "kernel_ms": dev_ms, "bytes_per_unit": bytes_per_unit}
    if args.workload == "queries":
        # host-driven search: no single kernel bounds it; report the search's own counters
        r = last["r"]
        stt = r["stats"]
        nq = len(r["status"])
        out["roofline"] = None
        out["search"] = {
            "found": int((r["status"] == 0).sum()), "no_path": int((r["status"] == 1).sum()),
            "cap_reached": int((r["status"] == A.PP_ERR_LIMIT).sum()),
            "other_errors": int(((r["status"] < 0) & (r["status"] != A.PP_ERR_LIMIT)).sum()),
            "expansions": int(sum(stt[i].expansions for i in range(nq))),
            "poses_checked": int(sum(stt[i].poses_checked for i in range(nq))),
            "rs_paths": int(sum(stt[i].rs_paths for i in range(nq))),
            "t_heuristic_ms": float(stt[0].t_heuristic_ms), "t_search_ms": float(stt[0].t_search_ms),
            "note": "counters of rank 0's shard, last step"}
    if args.workload == "wavefront":
        out["config"]["relaxation"] = api.wavefront_stats(ctx)
        if world > 1:
            out["config"]["exchange_rounds"] = rounds[-1]

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        if args.workload == "collision":
            ox, oy = W.raster_to_points(occ)
            om = O.OracleMap(ox, oy)
            ns = args.cpu_sample or min(n, 1 << 23)  # ~10 s single-threaded
            t0 = time.perf_counter()
            st, ef = om.collision_flags(conf, x[:ns], y[:ns], yaw[:ns])
            dt = time.perf_counter() - t0
            got = flags[:ns].cpu().numpy()
            mism = int(((got & 1) != ef).sum())
            ncpu = min(os.cpu_count() or 1, 16)
            t0 = time.perf_counter()
            om.collision_flags(conf, x[:ns], y[:ns], yaw[:ns], threads=ncpu)
            dtm = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"first {ns} poses of the same batch, CPU oracle "
                                             "(is_collision restated, bucket-grid candidate filter)",
                                   "flag_mismatches_vs_gpu": mism,
                                   "all_cores": {"value": ns / dtm, "cores": ncpu}}
        elif args.workload == "rs":
            ns = args.cpu_sample or min(n, 1 << 18)
            t0 = time.perf_counter()
            O.rs_batch(q0[:ns], q1[:ns], maxc, threads=1)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"first {ns} pairs, CPU oracle generate_path + argmin"}
        elif args.workload == "expand":
            ns = args.cpu_sample or min(n, 1 << 15)
            ox, oy = W.raster_to_points(occ)
            om = O.OracleMap(ox, oy)
            t0 = time.perf_counter()
            st, eo = om.expand(conf, x[:ns], y[:ns], yaw[:ns], np.ones(ns, np.int64), np.zeros(ns), np.zeros(ns))
            dt = time.perf_counter() - t0
            gv = eout["valid"][: ns * P].cpu().numpy()
            mism = int((((gv & 1) != (eo["valid"] & 1)) & ((gv & 2) == 0)).sum())
            out["cpu_baseline"] = {"value": ns * P / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"first {ns} parents, CPU oracle calc_next_node",
                                   "valid_mismatches_vs_gpu": mism}
        elif args.workload == "analytic":
            ns = args.cpu_sample or min(n, 4096)
            ox, oy = W.raster_to_points(occ)
            om = O.OracleMap(ox, oy)
            t0 = time.perf_counter()
            st, efound, eidx, enp, ewords, ecost = om.analytic(conf, q0[:ns], q1[:ns])
            dt = time.perf_counter() - t0
            gfound, gidx = a_found[:ns].cpu().numpy(), a_idx[:ns].cpu().numpy()
            clean = (gfound & 2) == 0
            mism = int((((gfound & 1) != (efound & 1)) & clean).sum() +
                       (((gfound & 1) == 1) & ((efound & 1) == 1) & (gidx != eidx) & clean).sum())
            out["cpu_baseline"] = {"value": ns / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"first {ns} pairs, CPU oracle analystic_expansion",
                                   "mismatches_vs_gpu": mism,
                                   "found_fraction": float((efound & 1).mean())}
        elif args.workload == "queries":
            ns = args.cpu_sample or 16
            ox, oy = W.raster_to_points(occ)
            om = O.OracleMap(ox, oy)
            O.set_max_expansions(args.max_expansions)
            r = last["r"]
            t0 = time.perf_counter()
            mism = 0
            for i in range(ns):
                e = om.plan(conf, starts[i], goals[i])
                ok = e["status"] == r["status"][i] and e["stats"].expansions == r["stats"][i].expansions
                if ok and e["status"] == 0:
                    ok = abs(e["cost"] - r["cost"][i]) <= 1e-9 * abs(e["cost"]) and len(e["x"]) == r["len"][i]
                mism += (not ok) and r["stats"][i].ambiguous == 0
            dt = time.perf_counter() - t0
            O.set_max_expansions(0)
            out["cpu_baseline"] = {"value": ns / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"first {ns} queries of rank 0's shard, CPU oracle planner "
                                             "(heuristic recomputed per query as the reference does)",
                                   "mismatches_vs_gpu": int(mism)}
        else:
            ns = args.cpu_sample or min(ncell, 2048)
            sub = np.ascontiguousarray(obs_np[:ns, :ns]).copy()
            sub[0, :] = sub[-1, :] = 1
            sub[:, 0] = sub[:, -1] = 1
            sub[ns // 2, ns // 2] = 0
            t0 = time.perf_counter()
            O.wavefront(sub, ns // 2, ns // 2)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns * ns / dt, "unit": unit, "cores": 1, "kind": "port",
                                   "sample": f"{ns}x{ns} corner of the same raster, CPU oracle "
                                             "Dijkstra (dense arrays, binary heap)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
