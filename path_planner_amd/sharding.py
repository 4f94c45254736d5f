"""How the independent units of the hot path are dealt to the ranks of a multi-GPU job.

Poses (collision checks), start/goal pairs (Reeds-Shepp words) and planner queries are independent
of each other, so a rank works on its own share with no data-path collective; only results that a
caller wants in one place are gathered.  The synthetic streams are counter-based
(workloads.splitmix64), so rank r's share of a weak-scaling run is the slice
[r * n, (r + 1) * n) of the one stream a single rank would draw for world * n units.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import workloads as W


def pose_shard(n_per_rank, grid, rank):
    """(x, y, yaw) of this rank's n_per_rank poses."""
    return W.random_poses(n_per_rank, grid, start=rank * n_per_rank)


def pair_shard(n_per_rank, rank, extent=100.0):
    """(q0[n,3], q1[n,3]) of this rank's n_per_rank start/goal pairs."""
    return W.rs_pairs(n_per_rank, start=rank * n_per_rank, extent=extent)


def query_shard(ids, rank, world):
    """This rank's share of a fixed list of queries, dealt round-robin: long and short queries
    spread evenly, and so do the goals -- every rank then relaxes every goal's heuristic grid
    (see query_shard_by_goal for the deal that keeps goal groups together)."""
    return ids[rank::world]


def goal_order(goal_keys, world=1):
    """Positions of a query list ordered by goal (stable: queries of one goal keep their order), the
    goal groups taken in turn: group r of the sorted goals comes in the r mod world-th stretch of the
    order, so that a run of neighbouring (often equally hard) goals is spread over the ranks.
    goal_keys: one hashable-by-value row per query (e.g. the goal's heuristic-grid cell)."""
    keys = np.asarray(goal_keys)
    keys = keys.reshape(len(keys), -1)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    return np.lexsort((np.arange(len(inv)), inv, inv % world))


def query_shard_by_goal(goal_keys, rank, world):
    """Positions (into the undealt list) of this rank's share when the list is dealt by goal: the
    queries are ordered by goal (goal_order) and cut into `world` runs of equal length (+-1), so a
    rank holds whole goal groups except for at most one group shared with each neighbour.  The planner
    relaxes one heuristic grid per distinct goal cell of its batch: dealt this way a rank relaxes
    about n_goals / world + 1 grids instead of all of them, at equal query counts.  (Queries of
    different goals can differ in length, so the shares are equal in queries, not in work.)"""
    order = goal_order(goal_keys, world)
    n = len(order)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    return np.sort(order[lo:hi])


def gather_by_position(part, positions, rank, world, total, group=None):
    """Rank 0 gets the per-query rows back in the order of the undealt list, whatever the deal:
    every rank sends its rows with the positions they belong at.  Other ranks get None."""
    part = np.ascontiguousarray(part)
    positions = np.asarray(positions, dtype=np.int64)
    if world == 1:
        out = np.empty((total,) + part.shape[1:], dtype=part.dtype)
        out[positions] = part
        return out
    parts = [None] * world if rank == 0 else None
    dist.gather_object((positions, part), parts, dst=0, group=group)
    if rank != 0:
        return None
    out = np.empty((total,) + part.shape[1:], dtype=part.dtype)
    seen = np.zeros(total, dtype=bool)
    for pos, p in parts:
        assert not seen[pos].any(), "a query was dealt twice"
        seen[pos] = True
        out[pos] = p
    assert seen.all(), "a query was not dealt"
    return out


def gather_slices(part, rank, world, group=None):
    """Rank 0 gets the ranks' arrays (equal dtype and trailing shape) joined in rank order -- the
    inverse of pose_shard / pair_shard.  Other ranks get None."""
    part = np.ascontiguousarray(part)
    if world == 1:
        return part
    parts = [None] * world if rank == 0 else None
    dist.gather_object(part, parts, dst=0, group=group)
    return np.concatenate(parts, axis=0) if rank == 0 else None


def gather_dealt(part, rank, world, total, group=None):
    """Rank 0 gets the per-query rows back in the order of the undealt list -- the inverse of
    query_shard.  Other ranks get None."""
    part = np.ascontiguousarray(part)
    if world == 1:
        return part
    parts = [None] * world if rank == 0 else None
    dist.gather_object(part, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = np.empty((total,) + part.shape[1:], dtype=part.dtype)
    for r, p in enumerate(parts):
        out[r::world] = p
    return out


def sum_over_ranks(value, device, group=None):
    """All-reduce (SUM) of one number -- the only collective of a sharded run, for its totals."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
