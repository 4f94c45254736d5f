"""The analytic-expansion entry point that allocates nothing (pp_analytic_batch_dev_static) against
the entry point it is a variant of (pp_analytic_batch_dev): same inputs, bit-identical outputs, on
two calls in a row, the second with a smaller batch (the device array keeps the first call's
records), and a clear error above the stated maximum.  The property itself is checked too: on a
context that has never allocated, the call leaves the device's free memory where it was (the
allocating entry point, on the same context, lowers it), and it can be recorded into a HIP graph
in global capture mode, in which hipMalloc is refused."""
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
    # a single pair
    _same(_run(api.analytic_expansion_dev_static, ctx, m, s[:1], g[:1]),
          _run(api.analytic_expansion_dev, ctx, m, s[:1], g[:1]))
    # an empty batch: PP_OK, no output touched (the pointers are not even looked at)
    out0 = _outputs(4)
    st = ctx.lib.pp_analytic_batch_dev_static(ctx.h, m.h, C.byref(CONF), 0, None, None, None, None, None, None)
    assert st == A.PP_OK
    ctx.synchronize()
    assert (out0[1].cpu().numpy() == -7).all() and (out0[0].cpu().numpy() == 0xEE).all()
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


def _hip():
    return C.CDLL("libamdhip64.so")


def test_static_entry_point_allocates_nothing():
    """A context that has never reserved its scratch buffer: the static call must not make it do so
    (pp_ctx_device_bytes stays 0), its result equals the allocating form's, and once the stream's
    queue is warm (the runtime sets a queue up at its first launch) a call leaves the device's free
    memory where it was.  The allocating form on the same context shows both measurements can see a
    reservation."""
    import torch
    torch.cuda.init()
    occ = W.occupancy_grid(256, 0.01)
    s, g = W.rs_pairs(2048, extent=200.0)
    s[:, :2] = (s[:, :2] + 28.0) * 0.5
    g[:, :2] = (g[:, :2] + 28.0) * 0.5
    warm = api.Context(0)
    mw = api.ObstacleMap(warm, raster=occ)
    ref = _run(api.analytic_expansion_dev, warm, mw, s, g)
    assert warm.device_bytes() > 0
    ctx = api.Context(0)    # fresh: no scratch, no staging
    m = api.ObstacleMap(ctx, raster=occ)
    q6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([s.T, g.T]))).cuda()
    assert ctx.device_bytes() == 0
    out = _outputs(len(s))
    api.analytic_expansion_dev_static(ctx, m, CONF, q6, *out)   # first launch on this stream
    ctx.synchronize()
    assert ctx.device_bytes() == 0
    _same(tuple(t.cpu().numpy() for t in out), ref)
    out = _outputs(len(s))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    api.analytic_expansion_dev_static(ctx, m, CONF, q6, *out)
    ctx.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, "the static entry point changed the device's free memory by %d bytes" % (free0 - free1)
    assert ctx.device_bytes() == 0
    _same(tuple(t.cpu().numpy() for t in out), ref)
    # the allocating form on the same context does reserve memory, and both measurements see it
    out2 = _outputs(len(s))
    api.analytic_expansion_dev(ctx, m, CONF, q6, *out2)
    ctx.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 < free1 and ctx.device_bytes() > 0
    for o in (m, ctx, mw, warm):
        o.close()


def test_static_entry_point_under_stream_capture():
    """Recorded into a HIP graph on the context's stream (global capture mode, in which hipMalloc is
    refused) and replayed: same results as the direct call."""
    import torch
    hip = _hip()
    occ = W.occupancy_grid(256, 0.01)
    s, g = W.rs_pairs(1024, extent=200.0)
    s[:, :2] = (s[:, :2] + 28.0) * 0.5
    g[:, :2] = (g[:, :2] + 28.0) * 0.5
    ctx = api.Context(0)
    m = api.ObstacleMap(ctx, raster=occ)
    ref = _run(api.analytic_expansion_dev_static, ctx, m, s, g)   # also the symbol lookup, outside the capture
    q6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([s.T, g.T]))).cuda()
    out = _outputs(len(s))
    torch.cuda.synchronize()
    stream = C.c_void_p(ctx.lib.pp_ctx_stream(ctx.h))
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0      # hipStreamCaptureModeGlobal
    st = ctx.lib.pp_analytic_batch_dev_static(ctx.h, m.h, C.byref(CONF), len(s), q6.data_ptr(), out[0].data_ptr(),
                                              out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                              out[4].data_ptr())
    rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert st == A.PP_OK and rc == 0 and graph.value, (st, rc)
    # nothing ran yet: the outputs still hold their fill
    torch.cuda.synchronize()
    assert (out[1].cpu().numpy() == -7).all()
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    assert hip.hipGraphLaunch(gexec, stream) == 0
    ctx.synchronize()
    _same(tuple(t.cpu().numpy() for t in out), ref)
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    m.close()
    ctx.close()
