"""The analytic-expansion entry point that allocates nothing (pp_analytic_batch_dev_static) against
the entry point it is a variant of (pp_analytic_batch_dev): same inputs, bit-identical outputs, on
two calls in a row, the second with a smaller batch (the device array keeps the first call's
records), and a clear error above the stated maximum."""
import ctypes as C

import numpy as np
import pytest

from path_planner_amd import _abi as A
from path_planner_amd import api
from path_planner_amd import workloads as W

pytestmark = pytest.mark.gpu
CONF = A.demo_config()


def _outputs(n):
    import torch
    return (torch.full((n,), 0xEE, dtype=torch.uint8, device="cuda"),
            torch.full((n,), -7, dtype=torch.int32, device="cuda"),
            torch.full((n,), -7, dtype=torch.int32, device="cuda"),
            torch.full((n, C.sizeof(A.RsWord)), 0xEE, dtype=torch.uint8, device="cuda"),
            torch.full((n,), -7.0, dtype=torch.float64, device="cuda"))


def _run(fn, ctx, m, s, g):
    import torch
    n = len(s)
    q6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([s.T, g.T]))).cuda()
    out = _outputs(n)
    fn(ctx, m, CONF, q6, *out)
    ctx.synchronize()
    found, idx, npaths, words, cost = (t.cpu().numpy() for t in out)
    return found, idx, npaths, words, cost


def _same(a, b):
    fa, ia, na, wa, ca = a
    fb, ib, nb, wb, cb = b
    assert (fa == fb).all() and (ia == ib).all() and (na == nb).all()
    assert (ca.view(np.uint64) == cb.view(np.uint64)).all()
    hit = (fb & 1) == 1
    assert (wa[hit] == wb[hit]).all()
    return int(hit.sum())


def test_static_entry_point_equals_allocating_one_twice():
    ctx = api.Context(0)
    assert ctx.lib.pp_analytic_static_max() == 4096
    occ = W.occupancy_grid(256, 0.01)
    m = api.ObstacleMap(ctx, raster=occ)
    nmax = int(ctx.lib.pp_analytic_static_max())
    s, g = W.rs_pairs(nmax, extent=200.0)
    s[:, :2] = (s[:, :2] + 28.0) * 0.5
    g[:, :2] = (g[:, :2] + 28.0) * 0.5
    # first call: the full 4096 pairs
    ref = _run(api.analytic_expansion_dev, ctx, m, s, g)
    got = _run(api.analytic_expansion_dev_static, ctx, m, s, g)
    hits = _same(got, ref)
    assert 20 < hits < nmax - 20
    # second call, smaller and different pairs: stale records of the first call must not leak in
    k = 777
    s2, g2 = g[100:100 + k][::-1].copy(), s[100:100 + k][::-1].copy()
    ref2 = _run(api.analytic_expansion_dev, ctx, m, s2, g2)
    got2 = _run(api.analytic_expansion_dev_static, ctx, m, s2, g2)
    _same(got2, ref2)
    # a single pair, and an empty batch
    _same(_run(api.analytic_expansion_dev_static, ctx, m, s[:1], g[:1]),
          _run(api.analytic_expansion_dev, ctx, m, s[:1], g[:1]))
    # above the maximum: refused, nothing launched
    import torch
    q6 = torch.zeros((6, nmax + 1), dtype=torch.float64, device="cuda")
    out = _outputs(nmax + 1)
    with pytest.raises(api.PlannerError) as e:
        api.analytic_expansion_dev_static(ctx, m, CONF, q6, *out)
    assert e.value.status == A.PP_ERR_CAPACITY
    ctx.synchronize()
    assert (out[1].cpu().numpy() == -7).all()
    m.close()
    ctx.close()
