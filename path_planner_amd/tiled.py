"""Holonomic heuristic grid tiled over several GPUs (one process per GPU).

The grid cost[x, y] (x-major, as the reference's hmap[x][y], astar.rs:194-196) is cut into
contiguous slabs of x-columns, one per rank.  Every rank relaxes its slab to the local fixpoint
(pp_wavefront_slab_dev), then neighbours swap their boundary columns and all ranks agree on
whether any boundary moved; the loop ends when none did.  The fixpoint d[v] = min_n fl(d[n] + w)
is unique and schedule-independent, so the assembled grid equals the single-device result (and
the reference's Dijkstra, astar.rs:153-198) bit for bit for any number of ranks.

The exchange is torch.distributed point-to-point (RCCL over xGMI under the "nccl" backend): two
yw-double columns per interior boundary per round, plus one int all-reduce.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi as A
from .api import PlannerError


def slab_bounds(xw, world, rank):
    """Columns [x_begin, x_end) owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(xw, world)
    x_begin = rank * base + min(rank, rem)
    return x_begin, x_begin + base + (1 if rank < rem else 0)


def make_slab(obs_full_or_fn, xw, yw, x_begin, x_end, device):
    """Obstacle slab with one halo column each side (blocked outside the grid) and an all-inf
    cost slab of the same shape.  `obs_full_or_fn` is a (xw, yw) uint8 array-like or a callable
    (x0, x1) -> uint8 tensor of columns [x0, x1)."""
    ncols = x_end - x_begin + 2
    obs = torch.ones((ncols, yw), dtype=torch.uint8)
    lo, hi = max(x_begin - 1, 0), min(x_end + 1, xw)
    if callable(obs_full_or_fn):
        part = obs_full_or_fn(lo, hi)
    else:
        part = torch.as_tensor(obs_full_or_fn[lo:hi])
    obs[lo - (x_begin - 1): hi - (x_begin - 1)] = part
    cost = torch.full((ncols, yw), float("inf"), dtype=torch.float64)
    return obs.to(device), cost.to(device)


class DeviceSlabSolver:
    """relax() on the GPU through the C ABI."""

    def __init__(self, ctx, xw, yw, x_begin, x_end, obs_slab, cost_slab, goal):
        self.ctx, self.xw, self.yw = ctx, xw, yw
        self.x_begin, self.x_end = x_begin, x_end
        self.obs, self.cost = obs_slab, cost_slab
        self.goal = goal
        self.iters = 0
        assert obs_slab.is_cuda and cost_slab.is_cuda and cost_slab.is_contiguous()
        assert tuple(cost_slab.shape) == (x_end - x_begin + 2, yw) == tuple(obs_slab.shape)

    def relax(self, seed_goal, halo_lo, halo_hi):
        ch_lo, ch_hi, it = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        st = self.ctx.lib.pp_wavefront_slab_dev(
            self.ctx.h, self.xw, self.yw, self.x_begin, self.x_end, self.obs.data_ptr(),
            self.cost.data_ptr(), 1 if seed_goal else 0, self.goal[0], self.goal[1],
            halo_lo.data_ptr() if halo_lo is not None else None,
            halo_hi.data_ptr() if halo_hi is not None else None,
            C.byref(ch_lo), C.byref(ch_hi), C.byref(it))
        if st != A.PP_OK:
            raise PlannerError(st, "pp_wavefront_slab_dev")
        self.iters += it.value
        return bool(ch_lo.value), bool(ch_hi.value)


def tiled_wavefront(solver, rank, world, group=None, max_rounds=1 << 20):
    """Drive `solver` (relax(seed_goal, halo_lo, halo_hi) -> (changed_lo, changed_hi) on
    solver.cost) to the global fixpoint.  Returns the number of exchange rounds.

    With the gloo backend the columns travel through host memory; with nccl they stay on the
    device."""
    cost = solver.cost
    on_host = world > 1 and dist.get_backend(group) == "gloo"
    xdev = torch.device("cpu") if on_host else cost.device
    yw = cost.shape[1]
    halo_lo = halo_hi = None
    recv_lo = torch.empty(yw, dtype=torch.float64, device=xdev) if rank > 0 else None
    recv_hi = torch.empty(yw, dtype=torch.float64, device=xdev) if rank < world - 1 else None
    rounds = 0
    seed = True
    while rounds < max_rounds:
        ch_lo, ch_hi = solver.relax(seed, halo_lo, halo_hi)
        ch_lo, ch_hi = ch_lo and rank > 0, ch_hi and rank < world - 1  # only shared boundaries
        seed = False
        rounds += 1
        if world == 1:
            break
        if cost.is_cuda:
            torch.cuda.synchronize(cost.device)
        ops = []
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, cost[1].to(xdev).contiguous(), rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, recv_lo, rank - 1, group))
        if rank < world - 1:
            ops.append(dist.P2POp(dist.isend, cost[-2].to(xdev).contiguous(), rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, recv_hi, rank + 1, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        flag = torch.tensor([1 if (ch_lo or ch_hi) else 0], dtype=torch.int32, device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            break
        halo_lo = recv_lo.to(cost.device) if recv_lo is not None else None
        halo_hi = recv_hi.to(cost.device) if recv_hi is not None else None
    return rounds


def gather_grid(cost_slab, rank, world, xw, group=None):
    """Assemble the full (xw, yw) grid on rank 0 from the owned columns of every rank."""
    own = cost_slab[1:-1].cpu().contiguous()
    if world == 1:
        return own
    parts = [None] * world if rank == 0 else None
    dist.gather_object(own, parts, dst=0, group=group)
    if rank != 0:
        return None
    full = torch.cat(parts, dim=0)
    assert full.shape[0] == xw
    return full
