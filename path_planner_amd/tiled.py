"""Holonomic heuristic grid tiled over several GPUs (one process per GPU).

The grid cost[x, y] (x-major, as the reference's hmap[x][y], astar.rs:194-196) is cut into
contiguous slabs of x-columns, one per rank.  A rank keeps its slab on the device in the
relaxation's working layout (pp_slab_create) and the ranks advance together in cost order: every
exchange round each rank relaxes what it has pending up to `limit` (pp_slab_relax), neighbours swap
their boundary columns, and one MIN all-reduce of the ranks' lowest pending values gives the next
limit (lowest + band).  A slab therefore never relaxes from a neighbour's provisional values far
ahead of the front, and the loop ends when nothing is pending anywhere.  The fixpoint
d[v] = min_n fl(d[n] + w) is unique and schedule-independent, so the assembled grid equals the
single-device result (and the reference's Dijkstra, astar.rs:153-198) bit for bit for any number
of ranks and any band.

The exchange is torch.distributed point-to-point (RCCL over xGMI under the "nccl" backend): two
yw-double columns per interior boundary per round from buffers allocated once, plus the one-element
all-reduce issued alongside them; its result is the round's only host read.
"""
import ctypes as C
import math

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
    """A slab session on the GPU through the C ABI.

    step(halo_lo, halo_hi, limit) takes the neighbours' columns (device tensors or None), relaxes up
    to `limit`, refreshes self.edge_lo / self.edge_hi (this rank's first / last owned column, device
    tensors allocated once) and returns the lowest value still pending here or not yet seen by a
    neighbour (+inf: nothing)."""

    def __init__(self, ctx, xw, yw, x_begin, x_end, obs_slab, goal):
        assert obs_slab.is_cuda and obs_slab.is_contiguous() and obs_slab.dtype == torch.uint8
        assert tuple(obs_slab.shape) == (x_end - x_begin + 2, yw)
        self.ctx, self.xw, self.yw = ctx, xw, yw
        self.x_begin, self.x_end = x_begin, x_end
        self.device = obs_slab.device
        self.iters = 0
        # a side with no neighbour has no edge buffer: its values are nobody's halo
        self.edge_lo = self.edge_hi = None
        if x_begin > 0:
            self.edge_lo = torch.full((yw,), float("inf"), dtype=torch.float64, device=self.device)
        if x_end < xw:
            self.edge_hi = torch.full((yw,), float("inf"), dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)  # the library works on its own stream
        h = C.c_void_p()
        st = ctx.lib.pp_slab_create(ctx.h, xw, yw, x_begin, x_end, obs_slab.data_ptr(), goal[0], goal[1],
                                    C.byref(h))
        if st != A.PP_OK:
            raise PlannerError(st, "pp_slab_create")
        self.h = h

    def step(self, halo_lo, halo_hi, limit):
        pending, it = C.c_double(0.0), C.c_uint64(0)
        st = self.ctx.lib.pp_slab_relax(
            self.h, halo_lo.data_ptr() if halo_lo is not None else None,
            halo_hi.data_ptr() if halo_hi is not None else None, float(limit),
            self.edge_lo.data_ptr() if self.edge_lo is not None else None,
            self.edge_hi.data_ptr() if self.edge_hi is not None else None, C.byref(pending), C.byref(it))
        if st != A.PP_OK:
            raise PlannerError(st, "pp_slab_relax")
        self.iters += it.value
        return pending.value

    def reset(self, obs_slab):
        """Start over (all costs +inf, goal seeded) on a raster of the same shape."""
        assert obs_slab.is_cuda and obs_slab.is_contiguous() and tuple(obs_slab.shape) == (self.x_end - self.x_begin + 2, self.yw)
        for e in (self.edge_lo, self.edge_hi):
            if e is not None:
                e.fill_(float("inf"))
        torch.cuda.synchronize(self.device)
        st = self.ctx.lib.pp_slab_reset(self.h, obs_slab.data_ptr())
        if st != A.PP_OK:
            raise PlannerError(st, "pp_slab_reset")

    def read(self, out=None):
        """The dense slab (x_end - x_begin + 2 columns, halo columns included) as a device tensor."""
        shape = (self.x_end - self.x_begin + 2, self.yw)
        cost = out if out is not None else torch.empty(shape, dtype=torch.float64, device=self.device)
        assert tuple(cost.shape) == shape and cost.is_contiguous() and cost.dtype == torch.float64
        torch.cuda.synchronize(self.device)
        st = self.ctx.lib.pp_slab_read(self.h, cost.data_ptr())
        if st != A.PP_OK:
            raise PlannerError(st, "pp_slab_read")
        return cost

    def gather(self, x, y):
        """Costs of the global cells (x[i], y[i]) (device int64 / uint64 tensors) straight from the
        working copy."""
        assert x.is_cuda and y.is_cuda and x.shape == y.shape and x.dtype in (torch.int64, torch.uint64)
        x, y = x.contiguous(), y.contiguous()
        out = torch.empty(x.shape, dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)
        st = self.ctx.lib.pp_slab_gather(self.h, x.numel(), x.data_ptr(), y.data_ptr(), out.data_ptr())
        if st != A.PP_OK:
            raise PlannerError(st, "pp_slab_gather")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.pp_slab_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def tiled_wavefront(solver, rank, world, group=None, band=1024.0, max_rounds=1 << 20):
    """Drive `solver` (step(halo_lo, halo_hi, limit) -> lowest pending value, refreshing
    solver.edge_lo / solver.edge_hi) to the global fixpoint.  Returns the number of exchange rounds.

    With the gloo backend the columns travel through host memory (staging buffers allocated once);
    with nccl they stay on the device."""
    inf = float("inf")
    if world == 1:
        pending = solver.step(None, None, inf)
        assert pending == inf
        return 1
    dev, yw = solver.device, solver.yw
    on_host = dist.get_backend(group) == "gloo" and dev.type != "cpu"
    xdev = torch.device("cpu") if on_host else dev

    def buf():
        return torch.empty(yw, dtype=torch.float64, device=xdev)
    recv_lo, recv_hi = (buf() if rank > 0 else None), (buf() if rank < world - 1 else None)
    send_lo = (buf() if on_host else solver.edge_lo) if rank > 0 else None
    send_hi = (buf() if on_host else solver.edge_hi) if rank < world - 1 else None
    halo_lo_dev = torch.empty(yw, dtype=torch.float64, device=dev) if on_host and rank > 0 else recv_lo
    halo_hi_dev = torch.empty(yw, dtype=torch.float64, device=dev) if on_host and rank < world - 1 else recv_hi
    key = torch.empty(1, dtype=torch.float64, device=xdev)
    ops = []
    if rank > 0:
        ops += [dist.P2POp(dist.isend, send_lo, rank - 1, group), dist.P2POp(dist.irecv, recv_lo, rank - 1, group)]
    if rank < world - 1:
        ops += [dist.P2POp(dist.isend, send_hi, rank + 1, group), dist.P2POp(dist.irecv, recv_hi, rank + 1, group)]
    halo_lo = halo_hi = None
    limit = band  # the goal's cost is 0
    rounds = 0
    while rounds < max_rounds:
        pending = solver.step(halo_lo, halo_hi, limit)  # returns with the edge buffers written
        rounds += 1
        if on_host:
            if send_lo is not None:
                send_lo.copy_(solver.edge_lo)
            if send_hi is not None:
                send_hi.copy_(solver.edge_hi)
        key[0] = pending
        red = dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group, async_op=True)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        red.wait()
        lowest = float(key.item())  # the round's host read
        if lowest == inf:
            break
        limit = lowest + band
        if on_host:
            if recv_lo is not None:
                halo_lo_dev.copy_(recv_lo)
            if recv_hi is not None:
                halo_hi_dev.copy_(recv_hi)
            torch.cuda.synchronize(dev)  # the library reads the columns on its own stream
        halo_lo, halo_hi = (halo_lo_dev if rank > 0 else None), (halo_hi_dev if rank < world - 1 else None)
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
