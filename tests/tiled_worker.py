"""One rank of the tiled-heuristic test (launched by torch.distributed.run).

--mode host  : a heap-based slab relaxer in numpy/python stands in for the device kernel, so the
               exchange protocol of path_planner_amd.tiled is exercised on CPU with gloo.
--mode device: the slab kernel through the C ABI; every rank uses cuda:0 (one-GPU box), the
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
    """Test stand-in for pp_wavefront_slab_dev with the same contract."""

    def __init__(self, xw, yw, x_begin, x_end, obs_slab, cost_slab, goal):
        self.xw, self.yw, self.x_begin, self.x_end = xw, yw, x_begin, x_end
        self.obs, self.cost, self.goal = obs_slab.numpy(), cost_slab, goal

    def relax(self, seed_goal, halo_lo, halo_hi):
        c = self.cost.numpy()
        ncols = c.shape[0]
        heap = []
        for col, h in ((0, halo_lo), (ncols - 1, halo_hi)):
            if h is None:
                continue
            h = h.numpy()
            for y in np.nonzero(h < c[col])[0]:
                c[col, y] = h[y]
                heap.append((float(h[y]), col, int(y)))
        gx, gy = self.goal
        if seed_goal and self.x_begin <= gx < self.x_end:
            c[gx - self.x_begin + 1, gy] = 0.0
            heap.append((0.0, gx - self.x_begin + 1, gy))
        heapq.heapify(heap)
        diag = math.hypot(1.0, 1.0)
        ch_lo = ch_hi = False
        while heap:
            d, lx, y = heapq.heappop(heap)
            if d != c[lx, y]:
                continue
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
                        ch_lo |= nlx == 1
                        ch_hi |= nlx == ncols - 2
        return ch_lo, ch_hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="host")
    ap.add_argument("--xw", type=int, default=61)
    ap.add_argument("--yw", type=int, default=47)
    ap.add_argument("--density", type=float, default=0.25)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--goal", type=int, nargs=2, default=None)
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
        obs_s, cost_s = tiled.make_slab(obs, args.xw, args.yw, x0, x1, dev)
        solver = tiled.DeviceSlabSolver(ctx, args.xw, args.yw, x0, x1, obs_s, cost_s, goal)
    else:
        obs_s, cost_s = tiled.make_slab(obs, args.xw, args.yw, x0, x1, "cpu")
        solver = HostSlabSolver(args.xw, args.yw, x0, x1, obs_s, cost_s, goal)
    rounds = tiled.tiled_wavefront(solver, rank, world)
    full = tiled.gather_grid(cost_s, rank, world, args.xw)
    if rank == 0:
        import oracle_lib as O
        st, exp = O.wavefront(obs, goal[0], goal[1])
        assert st == 0
        got = full.numpy()
        same = (got.view(np.uint64) == exp.view(np.uint64)).all()
        print(f"rounds={rounds} finite={int(np.isfinite(exp).sum())} {'OK' if same else 'MISMATCH'}",
              flush=True)
        if not same:
            sys.exit(1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
