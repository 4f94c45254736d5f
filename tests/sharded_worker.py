"""One rank of the sharded-units test (launched by torch.distributed.run, gloo).

Poses, Reeds-Shepp pairs and planner queries are dealt to the ranks by path_planner_amd.sharding --
the functions bench.py uses -- worked on independently and gathered on rank 0, which also works the
undealt list alone and compares the two bit for bit.

--mode host  : the CPU oracle stands in for the kernels, so dealing and gathering are exercised
               without a GPU.
--mode device: the kernels through the C ABI; every rank uses cuda:0 (one-GPU box).
"""
import argparse
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from path_planner_amd import _abi as A  # noqa: E402
from path_planner_amd import sharding as S  # noqa: E402
from path_planner_amd import workloads as W  # noqa: E402

CONF = A.demo_config()


class HostEngine:
    def __init__(self, occ):
        import oracle_lib as O
        self.O = O
        ox, oy = W.raster_to_points(occ)
        self.map = O.OracleMap(ox, oy)

    def flags(self, x, y, yaw):
        st, f = self.map.collision_flags(CONF, x, y, yaw)
        assert st == 0
        return f

    def words(self, q0, q1):
        st, npaths, best, words, _ = self.O.rs_batch(q0, q1, W.demo_maxc())
        assert st == 0
        return np.concatenate([npaths[:, None].astype(np.uint8), best[:, None].astype(np.uint8),
                               np.frombuffer(words, dtype=np.uint8).reshape(len(q0), -1)], axis=1)

    def plans(self, starts, goals):
        rows = []
        self.O.set_max_expansions(400)
        for s, g in zip(starts, goals):
            r = self.map.plan(CONF, s, g, cap=4096, closed_cap=1)
            rows.append(plan_row(r["status"], r["cost"], r["x"], r["y"], r["yaw"], r["stats"].expansions))
        self.O.set_max_expansions(0)
        return np.array(rows)


class DeviceEngine:
    def __init__(self, occ):
        from path_planner_amd import api
        self.api = api
        self.ctx = api.Context(0)
        self.map = api.ObstacleMap(self.ctx, raster=occ)

    def flags(self, x, y, yaw):
        return self.api.collision_flags(self.ctx, self.map, CONF, x, y, yaw)

    def words(self, q0, q1):
        npaths, best, words, _ = self.api.rs_batch(self.ctx, q0, q1, W.demo_maxc())
        return np.concatenate([npaths[:, None].astype(np.uint8), best[:, None].astype(np.uint8),
                               np.frombuffer(words, dtype=np.uint8).reshape(len(q0), -1)], axis=1)

    def plans(self, starts, goals):
        self.ctx.set_max_expansions(400)
        r = self.api.plan_batch(self.ctx, self.map, CONF, starts, goals, cap=4096)
        self.ctx.set_max_expansions(100000)
        return np.array([plan_row(int(r["status"][i]), r["cost"][i], r["x"][i, :r["len"][i]],
                                  r["y"][i, :r["len"][i]], r["yaw"][i, :r["len"][i]],
                                  r["stats"][i].expansions) for i in range(len(starts))])


def plan_row(status, cost, x, y, yaw, expansions):
    ok = status == 0
    return [float(status), float(cost) if ok else 0.0, float(len(x)) if ok else 0.0, float(expansions),
            float(np.sum(x)) if ok else 0.0, float(np.sum(y)) if ok else 0.0, float(np.sum(yaw)) if ok else 0.0]


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="host")
    ap.add_argument("--poses", type=int, default=3000)
    ap.add_argument("--pairs", type=int, default=400)
    ap.add_argument("--queries", type=int, default=13)   # odd: the ranks' shares differ in size
    ap.add_argument("--grid", type=int, default=256)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    occ = W.plan_map(args.grid, 0.006)
    eng = DeviceEngine(occ) if args.mode == "device" else HostEngine(occ)
    report = []

    # poses: rank r checks the slice [r n, (r + 1) n) of the stream
    x, y, yaw = S.pose_shard(args.poses, args.grid, rank)
    got = S.gather_slices(eng.flags(x, y, yaw), rank, world)
    if rank == 0:
        ax, ay, ayaw = W.random_poses(world * args.poses, args.grid)
        exp = eng.flags(ax, ay, ayaw)
        report.append(("poses", same_bits(got, exp), int(exp.sum())))

    # Reeds-Shepp pairs: same slicing
    q0, q1 = S.pair_shard(args.pairs, rank)
    got = S.gather_slices(eng.words(q0, q1), rank, world)
    if rank == 0:
        a0, a1 = W.rs_pairs(world * args.pairs)
        exp = eng.words(a0, a1)
        report.append(("pairs", same_bits(got, exp), int(exp[:, 0].sum())))

    # queries: collision-free candidates dealt round-robin
    cs, cg = W.plan_queries(6 * args.queries, args.grid, 4, min_dist=10.0, max_dist=60.0)
    inside = (cs[:, 0] > 8) & (cs[:, 0] < args.grid - 9) & (cs[:, 1] > 8) & (cs[:, 1] < args.grid - 9)
    free = (eng.flags(cs[:, 0], cs[:, 1], cs[:, 2]) == 0) & (eng.flags(cg[:, 0], cg[:, 1], cg[:, 2]) == 0)
    keep = np.nonzero(inside & free)[0][: args.queries]
    assert len(keep) == args.queries
    mine = S.query_shard(keep, rank, world)
    got = S.gather_dealt(eng.plans(cs[mine], cg[mine]), rank, world, len(keep))
    if rank == 0:
        exp = eng.plans(cs[keep], cg[keep])
        report.append(("queries", same_bits(got, exp), int((exp[:, 0] == 0).sum())))

    # the same list dealt by goal group (a rank then needs the heuristic grids of its goals only)
    cells = np.round(cg[keep][:, :2] / CONF.xy_reso).astype(np.int64)
    pos = S.query_shard_by_goal(cells, rank, world)
    mine_g = keep[pos]
    got = S.gather_by_position(eng.plans(cs[mine_g], cg[mine_g]), pos, rank, world, len(keep))
    my_goals = len({tuple(c) for c in cells[pos]})
    all_goals = len({tuple(c) for c in cells})
    fewer = S.sum_over_ranks(my_goals, "cpu")
    if rank == 0:
        # at most one goal group is shared with each neighbour
        report.append(("by_goal", same_bits(got, exp) and fewer <= all_goals + world - 1, int(fewer)))

    # the totals of a sharded run: one all-reduce
    total = S.sum_over_ranks(len(mine), "cpu")
    if rank == 0:
        report.append(("total", total == len(keep), int(total)))
        ok = all(r[1] for r in report)
        print(" ".join(f"{n}={'same' if s else 'DIFFERENT'}({c})" for n, s, c in report), "OK" if ok else "MISMATCH",
              flush=True)
        if not ok:
            sys.exit(1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
