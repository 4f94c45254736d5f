"""One rank of the tiled-heuristic test (launched by torch.distributed.run).

--mode host  : a heap-based slab relaxer in numpy/python stands in for the device slab session, so
               the exchange protocol of path_planner_amd.tiled is exercised on CPU with gloo.
--mode device: slab sessions through the C ABI; every rank uses cuda:0 (one-GPU box), the
               exchange still runs over gloo.
Rank 0 compares the assembled grid with the CPU oracle's Dijkstra bit for bit and prints "OK".
"""
import argparse
import heapq
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from path_planner_amd import tiled  # noqa: E402


class HostSlabSolver:
    """Test stand-in for a slab session (pp_slab_create / pp_slab_relax / pp_slab_read) with the same
    contract: a label-correcting search whose heap lives between steps and stops at the limit."""

    def __init__(self, xw, yw, x_begin, x_end, obs_slab, goal):
        self.xw, self.yw, self.x_begin, self.x_end = xw, yw, x_begin, x_end
        self.obs = obs_slab.numpy()
        self.device = torch.device("cpu")
        self.cost = torch.full((x_end - x_begin + 2, yw), float("inf"), dtype=torch.float64)
        self.edge_lo = torch.full((yw,), float("inf"), dtype=torch.float64) if x_begin > 0 else None
        self.edge_hi = torch.full((yw,), float("inf"), dtype=torch.float64) if x_end < xw else None
        self.heap = []
        gx, gy = goal
        if x_begin <= gx < x_end:
            self.cost[gx - x_begin + 1, gy] = 0.0
            self.heap.append((0.0, gx - x_begin + 1, gy))
        self.steps = []

    def step(self, halo_lo, halo_hi, limit):
        c = self.cost.numpy()
        ncols = c.shape[0]
        heap = self.heap
        for col, h in ((0, halo_lo), (ncols - 1, halo_hi)):
            if h is None:
                continue
            h = h.numpy()
            for y in np.nonzero(h < c[col])[0]:
                c[col, y] = h[y]
                heapq.heappush(heap, (float(h[y]), col, int(y)))
        diag = math.hypot(1.0, 1.0)
        popped = 0
        while heap and heap[0][0] <= limit:
            d, lx, y = heapq.heappop(heap)
            if d != c[lx, y]:
                continue
            popped += 1
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if dx == 0 and dy == 0:
                        continue
                    nlx, ny = lx + dx, y + dy
                    ngx = nlx + self.x_begin - 1
                    if not (1 <= nlx < ncols - 1 and 0 < ngx < self.xw and 0 < ny < self.yw):
                        continue
                    if self.obs[nlx, ny]:
                        continue
                    nd = d + (diag if dx and dy else 1.0)
                    if nd < c[nlx, ny]:
                        c[nlx, ny] = nd
                        heapq.heappush(heap, (nd, nlx, ny))
        self.steps.append(popped)
        pending = heap[0][0] if heap else float("inf")
        for edge, col in ((self.edge_lo, 1), (self.edge_hi, ncols - 2)):
            if edge is None:
                continue
            e = edge.numpy()
            moved = c[col] < e
            if moved.any():
                pending = min(pending, float(c[col][moved].min()))
                e[moved] = c[col][moved]
        return pending

    def read(self):
        return self.cost


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="host")
    ap.add_argument("--xw", type=int, default=61)
    ap.add_argument("--yw", type=int, default=47)
    ap.add_argument("--density", type=float, default=0.25)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--goal", type=int, nargs=2, default=None)
    ap.add_argument("--band", type=float, default=8.0)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rng = np.random.default_rng(args.seed)
    obs = (rng.random((args.xw, args.yw)) < args.density).astype(np.uint8)
    if args.density == 0.0:  # walls with gaps: long detours across slab boundaries
        for k, x in enumerate(range(4, args.xw - 2, 5)):
            obs[x, :] = 1
            obs[x, 2 if k % 2 else args.yw - 3] = 0
    goal = tuple(args.goal) if args.goal else (args.xw // 2, args.yw // 2)
    obs[goal] = 0
    x0, x1 = tiled.slab_bounds(args.xw, world, rank)
    if args.mode == "device":
        from path_planner_amd import api
        dev = torch.device("cuda", 0)
        ctx = api.Context(0)
        obs_s, _ = tiled.make_slab(obs, args.xw, args.yw, x0, x1, dev)
        solver = tiled.DeviceSlabSolver(ctx, args.xw, args.yw, x0, x1, obs_s, goal)
    else:
        obs_s, _ = tiled.make_slab(obs, args.xw, args.yw, x0, x1, "cpu")
        solver = HostSlabSolver(args.xw, args.yw, x0, x1, obs_s, goal)
    rounds = tiled.tiled_wavefront(solver, rank, world, band=args.band)
    full = tiled.gather_grid(solver.read(), rank, world, args.xw)
    if args.mode == "host":
        # cost ordering: no cell is settled twice as often as a run-to-fixpoint schedule would allow
        work = torch.tensor([float(sum(solver.steps))])
        dist.all_reduce(work)
    again = None
    if args.mode == "device":
        # the same session started over: buffers kept, result identical for another band
        solver.reset(obs_s)
        tiled.tiled_wavefront(solver, rank, world, band=4.0 * args.band + 3.0)
        again = tiled.gather_grid(solver.read(), rank, world, args.xw)
    if rank == 0:
        import oracle_lib as O
        st, exp = O.wavefront(obs, goal[0], goal[1])
        assert st == 0
        got = full.numpy()
        same = (got.view(np.uint64) == exp.view(np.uint64)).all()
        if again is not None:
            same = same and (again.numpy().view(np.uint64) == exp.view(np.uint64)).all()
        pops = f" pops={int(work.item())}" if args.mode == "host" else ""
        print(f"rounds={rounds} finite={int(np.isfinite(exp).sum())}{pops} {'OK' if same else 'MISMATCH'}",
              flush=True)
        if not same:
            sys.exit(1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
