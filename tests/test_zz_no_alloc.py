"""The entry point that allocates nothing (rt_render_shard_device_noalloc): for callers that cannot call
hipMalloc.  Its tile-time memory is a static device array of the library; the pixels must be those of
rt_render_shard_device.  Called through the C ABI twice in a row, the second time with a smaller frame,
so that what the first call left in the array must not matter."""
import ctypes as C

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu


def _shard(ctx, fn, w, h, **kw):
    import torch
    p = ctx.params(w, h, **kw)
    per, _ = ctx.shard_tiles(p)
    n = per * rt.TILE * rt.TILE * 3
    f = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    b = torch.full((n,), 77, dtype=torch.uint8, device="cuda")
    rc = fn(ctx.h, C.byref(p), C.c_void_p(f.data_ptr()), C.c_void_p(b.data_ptr()))
    assert rc == 0, ctx.L.rt_last_error(ctx.h).decode()
    ctx.sync()
    return f.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("preset", ["teapot", "shells"])
def test_no_alloc_entry_point_matches_the_allocating_one(preset, host_scene):
    L = rt.amd()
    ctx = rt.Context(0)
    try:
        ctx.upload(host_scene(preset))
        for (w, h, kw) in ((400, 240, dict(mode=rt.MODE_GRID, grid_m=3)),       # 25 x 15 tiles
                           (150, 90, dict(mode=rt.MODE_GRID, grid_m=2, fov=60.0)),  # smaller: 10 x 6 tiles, cut at both edges
                           (150, 90, dict(mode=rt.MODE_GRID, grid_m=2, fov=60.0)),  # again: now with tile times of its own
                           (208, 120, dict(mode=rt.MODE_REF_AA9, rank=1, world_size=3))):
            ref, ref8 = _shard(ctx, L.rt_render_shard_device, w, h, **kw)
            got, got8 = _shard(ctx, L.rt_render_shard_device_noalloc, w, h, **kw)
            assert np.array_equal(got8, ref8), (preset, w, h)
            assert np.array_equal(got, ref, equal_nan=True), (preset, w, h)
            assert ctx.counters().render_variant == rt.KERNEL_PER_TILE - 1
            assert ref8.std() > 3  # a picture
    finally:
        ctx.close()


def test_no_alloc_calls_of_two_contexts_do_not_disturb_each_other(host_scene):
    """Two contexts on one GPU, each on its own stream, call the entry point back to back without waiting for each
    other: they share the GPU's static array, the library orders their launches, and both get the pixels of the
    allocating entry point.  Free device memory does not shrink over the calls."""
    import torch
    L = rt.amd()
    a, b = rt.Context(0), rt.Context(0)
    try:
        a.upload(host_scene("teapot"))
        b.upload(host_scene("shells"))
        jobs = [(a, 320, 200, dict(mode=rt.MODE_GRID, grid_m=3)), (b, 208, 120, dict(mode=rt.MODE_GRID, grid_m=2))]
        want = [_shard(ctx, L.rt_render_shard_device, w, h, **kw) for ctx, w, h, kw in jobs]
        bufs = []
        for ctx, w, h, kw in jobs:
            p = ctx.params(w, h, **kw)
            per, _ = ctx.shard_tiles(p)
            n = per * rt.TILE * rt.TILE * 3
            bufs.append((p, torch.zeros(n, dtype=torch.float32, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(6):  # frames of the two contexts interleaved, nobody waits
            for (ctx, w, h, kw), (p, f, u8) in zip(jobs, bufs):
                rc = L.rt_render_shard_device_noalloc(ctx.h, C.byref(p), C.c_void_p(f.data_ptr()), C.c_void_p(u8.data_ptr()))
                assert rc == 0, ctx.L.rt_last_error(ctx.h).decode()
        a.sync(); b.sync()
        assert torch.cuda.mem_get_info()[0] >= free_before
        for (ref, ref8), (p, f, u8) in zip(want, bufs):
            assert np.array_equal(u8.cpu().numpy(), ref8)
            assert np.array_equal(f.cpu().numpy(), ref, equal_nan=True)
    finally:
        a.close(); b.close()


def test_no_alloc_entry_point_works_before_anything_was_allocated(host_scene):
    """A fresh context whose first render call is the entry point that allocates nothing (no tile memory of the
    context's own exists yet), with more tiles than a small frame would have made it allocate: free device memory is
    the same after the calls, and the pixels are the allocating entry point's."""
    import torch
    L = rt.amd()
    ctx = rt.Context(0)
    try:
        ctx.upload(host_scene("teapot"))
        w, h, kw = 1600, 1200, dict(mode=rt.MODE_REF_1SPP)  # 7500 tiles
        p = ctx.params(w, h, **kw)
        per, _ = ctx.shard_tiles(p)
        n = per * rt.TILE * rt.TILE * 3
        f = torch.zeros(n, dtype=torch.float32, device="cuda")
        u8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(2):
            assert L.rt_render_shard_device_noalloc(ctx.h, C.byref(p), C.c_void_p(f.data_ptr()), C.c_void_p(u8.data_ptr())) == 0
        ctx.sync()
        assert torch.cuda.mem_get_info()[0] == free_before
        ref, ref8 = _shard(ctx, L.rt_render_shard_device, w, h, **kw)
        assert np.array_equal(u8.cpu().numpy(), ref8) and np.array_equal(f.cpu().numpy(), ref, equal_nan=True)
    finally:
        ctx.close()


def test_no_alloc_entry_point_refuses_what_does_not_fit(host_scene):
    L = rt.amd()
    ctx = rt.Context(0)
    try:
        ctx.upload(host_scene("analytic"))
        p = ctx.params(16 * 300, 16 * 300, mode=rt.MODE_REF_1SPP)  # 90 000 tiles: more than the static array holds
        rc = L.rt_render_shard_device_noalloc(ctx.h, C.byref(p), None, None)
        assert rc == -5 and b"65536" in L.rt_last_error(ctx.h)  # RT_ERR_LIMIT, before anything is launched
        p = ctx.params(64, 64, count_work=True)
        assert L.rt_render_shard_device_noalloc(ctx.h, C.byref(p), None, None) == -1
    finally:
        ctx.close()
