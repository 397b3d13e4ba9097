"""Parity tests proper: the HIP kernels, called through the C ABI (ray_tracer_amd.Context forwards to
librt_amd.so), against the CPU oracle and the committed golden vectors.  Bars: tests/parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import parity
import ray_tracer_amd as rt
from ray_tracer_amd import tiling

pytestmark = pytest.mark.gpu
PRESETS = ["analytic", "teapot", "teapot_head", "shells", "textured"]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_native_library_is_what_runs(gpu_ctx):
    """The HIP extension must be the code that executes: it is loaded from the tree and a context exists."""
    gpu_ctx("analytic")
    maps = open("/proc/self/maps").read()
    assert os.path.join(rt.ROOT, "ray_tracer_amd", "librt_amd.so") in maps


@pytest.mark.parametrize("preset", PRESETS)
@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_BRUTE])
def test_golden_rays(preset, accel, gpu_ctx):
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    hits, rad = gpu_ctx(preset).trace_rays(g["o"], g["d"], max_depth=int(g["max_depth"]), accel=accel)
    parity.check_hits(g["hits"], hits, f"{preset}/accel{accel}")
    parity.check_radiance(g["radiance"], rad, f"{preset}/accel{accel}")


@pytest.mark.parametrize("preset", PRESETS)
def test_golden_frames(preset, gpu_ctx):
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    for mode, key in ((rt.MODE_REF_1SPP, "frame_1spp"), (rt.MODE_REF_AA9, "frame_aa9")):
        rgb, rgb8 = gpu_ctx(preset).render(64, 48, mode=mode, max_depth=int(g["max_depth"]))
        parity.check_radiance(g[key], rgb, f"{preset}/{key}")
        parity.check_rgb8(g[key + "_rgb8"], rgb8, f"{preset}/{key}")


@pytest.mark.parametrize("preset", PRESETS)
def test_live_oracle_rays(preset, gpu_ctx, oracle_scene, host_scene):
    """A different, seeded ray set checked against the oracle computed here."""
    osc = oracle_scene(preset)
    o, d = parity.ray_set(osc, host_scene(preset).flatten().contents, w=80, h=60, seed=5, n_random=1000)
    ref_hits = osc.find_closest_hit(o, d)
    ref_rad, ref_rays = osc.trace_rays(o, d, 8)
    ctx = gpu_ctx(preset)
    ctx.reset_counters()
    hits, rad = ctx.trace_rays(o, d, max_depth=8)
    parity.check_hits(ref_hits, hits, preset)
    parity.check_radiance(ref_rad, rad, preset)
    # ray accounting: one ray per closest-hit query, as the oracle counts them (ties may shift a few)
    assert abs(int(ctx.counters().rays) - int(ref_rays.sum())) <= 2e-3 * int(ref_rays.sum())


@pytest.mark.parametrize("preset", ["analytic", "teapot", "shells"])
def test_per_object_intersect(preset, gpu_ctx, oracle_scene, host_scene):
    osc = oracle_scene(preset)
    o, d = parity.ray_set(osc, host_scene(preset).flatten().contents, w=48, h=36, seed=6, n_random=500)
    for obj in range(osc.num_objects):
        ref = osc.intersect_object(obj, o, d)
        got = gpu_ctx(preset).intersect_object(obj, o, d)
        parity.check_hits(ref, got, f"{preset}/object{obj}")


@pytest.mark.parametrize("preset,mode,grid,size", [("analytic", rt.MODE_REF_AA9, 1, (160, 90)), ("teapot", rt.MODE_GRID, 4, (160, 90)),
                                                   ("shells", rt.MODE_GRID, 2, (96, 54)), ("textured", rt.MODE_GRID, 3, (128, 128)),
                                                   ("teapot", rt.MODE_REF_1SPP, 1, (250, 130))])
def test_frames_against_live_oracle(preset, mode, grid, size, gpu_ctx, oracle_scene):
    w, h = size  # includes sizes that are not multiples of the 16-pixel tile
    ref_rgb, ref_rgb8, ref_rays = oracle_scene(preset).render(w, h, mode=mode, grid_m=grid, max_depth=8)
    ctx = gpu_ctx(preset)
    ctx.reset_counters()
    rgb, rgb8 = ctx.render(w, h, mode=mode, grid_m=grid, max_depth=8)
    parity.check_radiance(ref_rgb, rgb, preset)
    parity.check_rgb8(ref_rgb8, rgb8, preset)
    assert abs(int(ctx.counters().rays) - ref_rays) <= 2e-3 * ref_rays


def test_lbvh_equals_brute_force_on_gpu(gpu_ctx):
    """Identical images from the LBVH and from the reference-order linear scan (bitwise: the tree only
    chooses which triangles are tested)."""
    for preset in ("teapot", "shells"):
        ctx = gpu_ctx(preset)
        a, a8 = ctx.render(192, 108, mode=rt.MODE_REF_1SPP, accel=rt.ACCEL_LBVH)
        b, b8 = ctx.render(192, 108, mode=rt.MODE_REF_1SPP, accel=rt.ACCEL_BRUTE)
        assert np.array_equal(a8, b8)
        assert np.array_equal(a, b, equal_nan=True)


def test_lbvh_structure_on_gpu(gpu_ctx, host_scene):
    for preset in ("teapot", "shells"):
        d = host_scene(preset).flatten().contents
        ctx = gpu_ctx(preset)
        pos_all = np.ctypeslib.as_array(d.tri_pos, shape=(d.n_triangles * 9,)).reshape(-1, 3, 3)
        for mi in range(d.n_meshes):
            m = d.meshes[mi]
            nodes, order = ctx.copy_bvh(mi)
            parity.check_lbvh(nodes, order, pos_all, m.tri_offset, m.tri_count)
        assert 0 < ctx.counters().bvh_max_depth <= 64


@pytest.mark.parametrize("builder", ["radix", "sah"])
def test_cpu_twin_and_gpu_agree_on_the_lbvh(builder, host_scene, monkeypatch):
    """The serial host build and the GPU build produce the same tree bit for bit (same keys, same sort, same
    splits, same node ids, same fit), with the surface-area splits and with the radix tree."""
    import twin_lib as T
    monkeypatch.setenv("RT_BUILD", builder)
    for preset in ("teapot", "shells"):  # shells: with triangles cut into pieces
        d = host_scene(preset).flatten().contents
        tw = T.Twin(host_scene(preset))
        ctx = rt.Context(0)
        try:
            ctx.upload(host_scene(preset))
            assert ctx.counters().bvh_max_depth == tw.height
            for mi in range(d.n_meshes):
                n0, o0 = tw.copy_bvh(mi)
                n1, o1 = ctx.copy_bvh(mi)
                assert np.array_equal(o0, o1)
                assert np.array_equal(n0.view(np.uint32), n1.view(np.uint32))
        finally:
            ctx.close()


def test_sharded_render_reassembles_bitwise(gpu_ctx):
    """Tile sharding as used across GPUs: every rank's shard rendered on this GPU, gathered in rank-major
    order and de-interleaved by rt_untile_device, equals the single-rank frame bit for bit."""
    import torch
    ctx = gpu_ctx("teapot")
    # 13 x 8 tiles: not divisible by the world sizes below; 200 wide: the last tile column sticks out and the
    # 8-bit rows are not 16-byte aligned (byte-wise de-interleave), 208 wide: whole tiles, 16-byte pieces
    for w, h in ((200, 120), (208, 120)):
        _sharded_render_case(ctx, w, h)


def _sharded_render_case(ctx, w, h):
    import torch
    full, full8 = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=2)
    for world in (2, 3, 8):
        p0 = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=2, world_size=world)
        per, total = ctx.shard_tiles(p0)
        assert total == 13 * 8 and per == -(-total // world)
        shards = torch.zeros((world, per, 16, 16, 3), dtype=torch.float32, device="cuda")
        shards8 = torch.zeros((world, per, 16, 16, 3), dtype=torch.uint8, device="cuda")
        for r in range(world):
            p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=2, rank=r, world_size=world)
            ctx.render_shard_device(p, shards[r].data_ptr(), shards8[r].data_ptr())
        frame = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        frame8 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        ctx.untile_device(p0, shards.data_ptr(), frame.data_ptr(), shards8.data_ptr(), frame8.data_ptr())
        ctx.sync()
        assert np.array_equal(frame.cpu().numpy(), full, equal_nan=True)
        assert np.array_equal(frame8.cpu().numpy(), full8)
        # the host-side restatement of the tile rule agrees with the kernels
        assert np.array_equal(tiling.frame_from_shards(shards.cpu().numpy(), w, h), full, equal_nan=True)
        # rt_render with world_size > 1 writes only this rank's tiles: the rest of the caller's buffers is left as it was
        part, part8 = np.full((h, w, 3), -7.0, np.float32), np.full((h, w, 3), 201, np.uint8)
        ctx.render(w, h, into=(part, part8), mode=rt.MODE_GRID, grid_m=2, rank=1, world_size=world)
        is_mine = tiling.frame_from_shards(np.stack([np.full_like(tiling.shard_from_frame(full, r, world), float(r == 1)) for r in range(world)]), w, h) == 1
        assert np.array_equal(part[is_mine], full[is_mine], equal_nan=True) and (part[~is_mine] == -7.0).all()
        assert np.array_equal(part8[is_mine], full8[is_mine]) and (part8[~is_mine] == 201).all()
        # ... so that the ranks of a frame can be composited into one host buffer, one after the other
        for r in range(world):
            if r != 1:
                ctx.render(w, h, into=(part, part8), mode=rt.MODE_GRID, grid_m=2, rank=r, world_size=world)
        assert np.array_equal(part, full, equal_nan=True) and np.array_equal(part8, full8)


def test_full_size_properties(gpu_ctx):
    """Size-independent checks at the benchmark's size (teapot, 3840x2160, 16 spp)."""
    ctx = gpu_ctx("teapot")
    w, h = 3840, 2160
    ctx.reset_counters()
    rgb, rgb8 = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=4, max_depth=8)
    c = ctx.counters()
    # closed room, two lights, no reflective material: every primary ray hits something and casts 2 shadow rays
    assert c.rays_primary == w * h * 16 and c.rays_shadow == 2 * c.rays_primary and c.rays_secondary == 0
    assert np.isfinite(rgb).all() and rgb.min() >= 0
    # idempotence
    rgb_b, rgb8_b = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=4, max_depth=8)
    assert np.array_equal(rgb8, rgb8_b) and np.array_equal(rgb, rgb_b)
    # the quantised image is the oracle's tone map + truncation of the linear image
    sel = np.random.default_rng(0).integers(0, w * h, size=200000)
    assert np.abs(O.tonemap_quantise(rgb.reshape(-1, 3)[sel]).astype(int) - rgb8.reshape(-1, 3)[sel].astype(int)).max() <= 1
    # a 16-spp pixel is the mean of its taps: compare a crop against the oracle (few seconds of CPU)
    x0, y0, cw, ch = 1320, 590, 64, 24  # across the teapot's left silhouette (x = 1353 on row 600)
    osc = O.Scene("teapot")
    ref_rgb, ref_rgb8, _ = osc.render(w, h, mode=2, grid_m=4, max_depth=8, rect=(x0, y0, x0 + cw, y0 + ch))
    parity.check_radiance(ref_rgb[y0:y0 + ch, x0:x0 + cw], rgb[y0:y0 + ch, x0:x0 + cw], "4K crop")
    parity.check_rgb8(ref_rgb8[y0:y0 + ch, x0:x0 + cw], rgb8[y0:y0 + ch, x0:x0 + cw], "4K crop")
    # both the wall and the (red) teapot are in the crop
    crop = rgb8[y0:y0 + ch, x0:x0 + cw].reshape(-1, 3).astype(int)
    reddish = (crop[:, 0] > crop[:, 1] + 60).mean()
    assert 0.1 < reddish < 0.9


def test_error_paths(host_scene):
    ctx = rt.Context(0)
    with pytest.raises(rt.RtError, match="no scene"):
        ctx.render(16, 16)
    ctx.upload(host_scene("teapot"), build=False)
    with pytest.raises(rt.RtError, match="rt_build_accel"):
        ctx.render(16, 16, accel=rt.ACCEL_LBVH)
    ctx.render(16, 16, accel=rt.ACCEL_BRUTE)  # the linear scan needs no tree
    ctx.build_accel()
    ctx.build_accel()  # idempotent
    with pytest.raises(rt.RtError):
        ctx.render(0, 16)
    with pytest.raises(rt.RtError):
        ctx.render(16, 16, max_depth=rt.MAX_DEPTH + 1)
    with pytest.raises(rt.RtError):
        ctx.intersect_object(99, np.zeros((1, 3)), np.ones((1, 3)))
    bad = rt.SceneDesc.from_buffer_copy(host_scene("teapot").flatten().contents)
    bad.n_materials = 0
    with pytest.raises(rt.RtError, match="material"):
        ctx.upload(C.pointer(bad))
    ctx.close()


def test_empty_and_tiny_meshes_on_gpu():
    tri = np.array([[0, 0, 5, 0, 1, 5, 1, 0, 5], [1, 0, 5, 0, 1, 5, 1, 1, 5]], np.float32)
    nrm = np.tile(np.array([0, 0, -1], np.float32), (2, 3))
    o = np.array([[0.2, 0.2, 0], [0.8, 0.8, 0], [0.5, 0.5, 0], [0.2, 0.2, 0], [5, 5, 0]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [np.nan, 0, 1], [0, 0, 1]], np.float32)
    for n in (0, 1, 2):
        hs = rt.HostScene()
        hs.add_mesh_tris(tri[:n], nrm[:n], rt.mat_identity(), rt.Material(diffuse=(1, 0, 0)), boxed=False)
        hs.add_light((0, 5, 0), 1.0)
        ctx = rt.Context(0)
        ctx.upload(hs)
        hits, rad = ctx.trace_rays(o, d, max_depth=4)
        want = [[0, 0, 0, 0, 0], [1, 0, 1, 0, 0], [1, 1, 1, 0, 0]][n]
        assert list(hits["hit"]) == want
        assert np.isfinite(rad).all()
        ctx.close()


def _posed_room(angle, light_y, reflective=False):
    """The teapot room with the teapot turned by `angle` about y and the first light at height light_y."""
    hs = rt.HostScene()
    red = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(1.0, 0.2, 0.2), specular=0.5, shininess=10.0)
    wall = rt.Material(reflectiveness=0.4) if reflective else rt.Material()
    M = rt.rotate(angle, (0, 1, 0), rt.translate((0, 1, 10)))
    hs.add_mesh_obj(os.path.join(rt.ASSETS, "teapot.obj"), M, red, boxed=True, flip_normals=True)
    hs.add_plane((0, -3, 0), (0, 1, 0), wall)
    hs.add_plane((0, 1, 30), (0, 0, -1), rt.Material())
    hs.add_plane((0, 2, 40), (0.3, 0.1, -1), rt.Material())  # a general (not axis-aligned) plane
    hs.add_light((10, light_y, -7), 0.5)
    hs.add_light((2, 7, 1), 0.2)
    return hs


def test_scene_update_equals_fresh_upload():
    """rt_scene_update (new pose, lights, materials; same triangles) renders exactly what a fresh upload
    of the new scene renders, without rebuilding the LBVH."""
    ctx = rt.Context(0)
    ctx.upload(_posed_room(0.0, 16))
    first, _ = ctx.render(160, 96, mode=rt.MODE_GRID, grid_m=2)
    build_ms = ctx.counters().last_build_ms
    for angle, ly, refl in [(0.7, 16, False), (2.1, 9, False), (3.5, 12, True)]:
        scene = _posed_room(angle, ly, refl)
        ctx.update(scene)
        got, got8 = ctx.render(160, 96, mode=rt.MODE_GRID, grid_m=2)
        assert ctx.counters().last_build_ms == build_ms  # no rebuild happened
        fresh = rt.Context(0)
        fresh.upload(scene)
        want, want8 = fresh.render(160, 96, mode=rt.MODE_GRID, grid_m=2)
        fresh.close()
        assert np.array_equal(got, want, equal_nan=True) and np.array_equal(got8, want8)
        assert not np.array_equal(got, first)
    # geometry that does not match the upload is refused
    other = rt.HostScene()
    other.add_plane((0, -3, 0), (0, 1, 0), rt.Material())
    other.add_light((0, 5, 0), 1.0)
    with pytest.raises(rt.RtError, match="differ"):
        ctx.update(other)
    ctx.close()
    empty = rt.Context(0)
    with pytest.raises(rt.RtError, match="before rt_scene_upload"):
        empty.update(_posed_room(0.0, 16))
    empty.close()


def _cull_scene(kind):
    """Scenes around the shadow-query cull: it may only skip queries that cannot find an occluder."""
    hs = rt.HostScene()
    red = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(1.0, 0.2, 0.2), specular=0.5, shininess=10.0)
    teapot = os.path.join(rt.ASSETS, "teapot.obj")
    n_meshes = 5 if kind == "five_meshes" else 2 if kind == "two_meshes" else 1
    for k in range(n_meshes):
        M = rt.scale((0.5, 0.5, 0.5), rt.translate((-6 + 3 * k, 0, 9 + k))) if n_meshes > 1 else rt.translate((0, 1, 10))
        hs.add_mesh_obj(teapot, M, red, boxed=(k % 2 == 0), flip_normals=True)
    hs.add_plane((0, -3, 0), (0, 1, 0), rt.Material())
    hs.add_plane((0, 1, 30), (0, 0, -1), rt.Material())
    hs.add_plane((-15, 1, 0), (1, 0, 0), rt.Material())
    if kind == "light_in_sphere":
        hs.add_light((0.5, 3.5, 10.5), 0.3)     # inside the teapot's bounding sphere: no cone
    elif kind == "light_behind_plane":
        hs.add_plane((0, 12, 0), (0, 1, 0), rt.Material())  # a shelf above the teapot, its front side up ...
        hs.add_light((3, 10, 8), 0.4)                       # ... and a light below it: the shelf can shadow this light
    else:
        hs.add_light((10, 16, -7), 0.5)
    hs.add_light((2, 7, 1), 0.2)
    return hs


@pytest.mark.parametrize("kind", ["room", "light_in_sphere", "light_behind_plane", "two_meshes", "five_meshes"])
def test_shadow_query_cull_changes_nothing(kind):
    """The frame kernel of direct-lighting scenes (nested loops, wave-wide skipping of shadow queries) against
    rt_trace_rays on the same camera rays (general ray loop, no such skipping): identical radiance."""
    w, h, m = 96, 64, 2
    ctx = rt.Context(0)
    ctx.upload(_cull_scene(kind))
    frame, _ = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=m)
    o, d, _wts = O.camera_rays(w, h, mode=rt.MODE_GRID, grid_m=m)
    _, rad = ctx.trace_rays(o, d, want_hits=False)
    lbvh_brute, _ = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=m, accel=rt.ACCEL_BRUTE)
    ctx.close()
    rad = rad.reshape(h * w, m * m, 3)
    acc = np.zeros((h * w, 3), np.float32)
    for k in range(m * m):
        acc = acc + rad[:, k] * np.float32(1.0)
    want = (acc / np.float32(m * m)).reshape(h, w, 3)
    assert np.array_equal(frame, want, equal_nan=True)
    assert np.array_equal(frame, lbvh_brute, equal_nan=True)
    lit = frame.reshape(-1, 3).sum(axis=1)
    assert (lit > 0).mean() > 0.5 and np.unique(np.round(lit, 3)).size > 50  # a real picture with shading and shadows


@pytest.mark.parametrize("preset,mode,grid", [("teapot", rt.MODE_GRID, 4), ("analytic", rt.MODE_REF_AA9, 1), ("analytic", rt.MODE_REF_1SPP, 1),
                                              ("textured", rt.MODE_GRID, 3), ("shells", rt.MODE_GRID, 8), ("teapot", rt.MODE_GRID, 7)])
@pytest.mark.parametrize("phases", ["1", "2"])
def test_tap_splitting_kernel_is_bit_identical(preset, mode, grid, phases, host_scene, monkeypatch):
    """k_render_split (a pixel's taps spread over the four waves of a block -- with two phases also over the
    halves of each wave -- and summed in tap order by one wave) against k_render (one lane per pixel): the same
    frame bit for bit, for tap counts that do and do not divide by four, fill less than one round (8 taps) or
    need several."""
    w, h = 88, 50  # not a multiple of the tile size
    frames = []
    monkeypatch.setenv("RT_SPLIT_PHASES", phases)
    for min_taps in ("1", "100000"):
        monkeypatch.setenv("RT_SPLIT_MIN_TAPS", min_taps)
        ctx = rt.Context(0)
        ctx.upload(host_scene(preset))
        frames.append(ctx.render(w, h, mode=mode, grid_m=grid, max_depth=6))
        ctx.reset_counters()
        ctx.render(w, h, mode=mode, grid_m=grid, max_depth=6, count_work=True)
        frames.append(ctx.counters().rays)
        ctx.close()
    (split, split8), split_rays, (plain, plain8), plain_rays = frames
    assert np.array_equal(split, plain, equal_nan=True) and np.array_equal(split8, plain8)
    assert split_rays == plain_rays
    assert plain8.std() > 5  # a picture, not a constant


@pytest.mark.parametrize("collective", [False, True])
def test_frame_pipeline_overlapped_frames_equal_serial(gpu_ctx, collective):
    """Frames queued back to back through the two-stream pipeline (render of frame k+1 overlapping the
    exchange and untile of frame k, two shard and frame buffers) equal the frames rendered one by one.
    Every frame has another camera, so a buffer that is reused too early shows.  collective=True sends
    the shards through a real (single-rank) RCCL all-gather."""
    import torch
    import torch.distributed as dist
    from ray_tracer_amd.frames import FramePipeline
    ctx = gpu_ctx("teapot")
    w, h = 400, 240
    fovs = [90.0, 60.0, 75.0, 50.0, 100.0, 40.0, 85.0]
    expect = [ctx.render(w, h, fov=f, mode=rt.MODE_GRID, grid_m=2) for f in fovs]
    own_group = False
    if collective and not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29600 + os.getpid() % 1000}", rank=0,
                                world_size=1, device_id=torch.device("cuda", 0))
        own_group = True
    try:
        pipe = FramePipeline(ctx, w, h, float_frame=True, force_collective=collective)
        got, pending = [], None
        for f in fovs + [None]:
            nxt = pipe.submit(ctx.params(w, h, fov=f, mode=rt.MODE_GRID, grid_m=2)) if f is not None else None
            if pending is not None:  # collect frame k while frame k+1 is in flight
                slot, done = pending
                done.synchronize()
                got.append((pipe.frame[slot].cpu().numpy().reshape(h, w, 3).copy(),
                            pipe.frame8[slot].cpu().numpy().reshape(h, w, 3).copy()))
            pending = nxt
        pipe.close()
    finally:
        if own_group:
            dist.destroy_process_group()
    assert len(got) == len(fovs)
    for (e, e8), (g, g8) in zip(expect, got):
        assert np.array_equal(e8, g8)
        assert np.array_equal(e, g, equal_nan=True)


@pytest.mark.parametrize("builder", ["radix", "sah"])
def test_tall_tree_equals_brute_force_on_gpu(builder, monkeypatch):
    """The tall-tree mesh (Morton chain + thousands of equal codes; 42 levels as a radix tree): packet traversal
    and linear scan give the same records and radiance, both match the CPU twin, and both builders give the
    twin's tree."""
    import twin_lib as T
    monkeypatch.setenv("RT_BUILD", builder)
    hs, o, d = parity.deep_tree_scene()
    ctx = rt.Context(0)
    try:
        ctx.upload(hs)
        height = ctx.counters().bvh_max_depth
        assert (30 if builder == "radix" else 1) <= height <= 64
        a, ra = ctx.trace_rays(o, d, 4, rt.ACCEL_LBVH)
        b, rb = ctx.trace_rays(o, d, 4, rt.ACCEL_BRUTE)
        nodes, order = ctx.copy_bvh(0)
    finally:
        ctx.close()
    assert a.tobytes() == b.tobytes() and np.array_equal(ra, rb, equal_nan=True)
    tw = T.Twin(hs)
    assert tw.height == height
    n0, o0 = tw.copy_bvh(0)
    assert np.array_equal(o0, order) and np.array_equal(n0.view(np.uint32), nodes.view(np.uint32))
    c, rc = tw.trace_rays(o, d, 4, rt.ACCEL_LBVH)
    parity.check_hits(c, a, "tall tree")
    parity.check_radiance(rc, ra, "tall tree")


def test_every_render_kernel_writes_the_same_pixels(gpu_ctx):
    """The render kernels differ in who renders a pixel and when, never in its value: each kernel named through
    rt_render_params.kernel, over several frames (the per-tile and the queue kernel plan a frame from the tiles'
    times in the frame before), gives the bytes of the one-lane-per-pixel kernel.  The automatic choice is a fixed
    rule (nothing is timed): the same kernel for every frame of this sequence."""
    ctx = gpu_ctx("teapot")
    kw = dict(fov=77.0, mode=rt.MODE_GRID, grid_m=3)
    ref, ref8 = ctx.render(333, 177, kernel=rt.KERNEL_PIXEL, **kw)
    assert ctx.counters().render_variant == rt.KERNEL_PIXEL - 1
    for kernel in (rt.KERNEL_SPLIT, rt.KERNEL_SPLIT2, rt.KERNEL_PER_TILE, rt.KERNEL_QUEUE, rt.KERNEL_AUTO):
        variants = []
        for _ in range(4):
            rgb, rgb8 = ctx.render(333, 177, kernel=kernel, **kw)
            variants.append(ctx.counters().render_variant)
            assert np.array_equal(rgb8, ref8) and np.array_equal(rgb, ref, equal_nan=True), (kernel, variants)
        expect = kernel - 1 if kernel != rt.KERNEL_AUTO else rt.KERNEL_QUEUE - 1  # a mesh scene, nine taps, the queue kernel
        assert variants == [expect] * 4
    # scenes without meshes, and single-tap frames, go to the one-lane-per-pixel kernel
    ctx.render(96, 64, mode=rt.MODE_REF_1SPP)
    assert ctx.counters().render_variant == rt.KERNEL_PIXEL - 1
    gpu_ctx("analytic").render(96, 64, mode=rt.MODE_REF_AA9)
    assert gpu_ctx("analytic").counters().render_variant == rt.KERNEL_PIXEL - 1


def test_queue_kernel_list_of_more_tiles_than_one_planning_batch(gpu_ctx):
    """A frame of 82 944 tiles: the planning kernel takes the tiles in batches of 32 768, and the list it leaves must hold
    every tile once, the second time round from recorded times (some tiles split).  Same bytes as the pixel kernel."""
    ctx = gpu_ctx("teapot")
    kw = dict(mode=rt.MODE_GRID, grid_m=2, want_rgb=False)
    _, ref8 = ctx.render(4608, 4608, kernel=rt.KERNEL_PIXEL, **kw)
    for _ in range(3):
        _, rgb8 = ctx.render(4608, 4608, kernel=rt.KERNEL_QUEUE, **kw)
        assert ctx.counters().render_variant == rt.KERNEL_QUEUE - 1
        assert np.array_equal(rgb8, ref8)


def test_first_frame_of_a_view_records_tile_times(gpu_ctx):
    """A whole 4K frame of a mesh scene on one GPU: the rule's first frame of a view is k_render's, which leaves the
    tiles' times; the frames after it are the queue kernel's; a new view starts over.  All the same bytes."""
    ctx = gpu_ctx("teapot")
    kw = dict(mode=rt.MODE_GRID, grid_m=2)
    _, ref8 = ctx.render(3840, 2160, want_rgb=False, kernel=rt.KERNEL_PIXEL, **kw)
    variants = []
    for _ in range(3):
        _, rgb8 = ctx.render(3840, 2160, want_rgb=False, **kw)
        variants.append(ctx.counters().render_variant)
        assert np.array_equal(rgb8, ref8)
    assert variants == [rt.KERNEL_PIXEL - 1, rt.KERNEL_QUEUE - 1, rt.KERNEL_QUEUE - 1]
    ctx.render(3840, 2160, want_rgb=False, fov=60.0, **kw)
    assert ctx.counters().render_variant == rt.KERNEL_PIXEL - 1


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_equals_oracle_on_gpu(seed):
    """Random mixes of every object kind, transformation and material (tests/random_scenes.py): hit records
    and radiance of rt_trace_rays against the oracle, with the tree and with the linear scan, and a small
    frame of rt_render against the oracle's."""
    import random_scenes
    sp, what = random_scenes.build(seed)
    o, d = random_scenes.rays(sp, seed)
    ref = sp.orc.find_closest_hit(o, d)
    ref_rad, _ = sp.orc.trace_rays(o, d, 6)
    ref_rgb, ref_rgb8, _ = sp.orc.render(96, 72, mode=rt.MODE_REF_AA9, max_depth=6)
    ctx = rt.Context(0)
    try:
        ctx.upload(sp.host)
        for accel in (rt.ACCEL_LBVH, rt.ACCEL_BRUTE):
            hits, rad = ctx.trace_rays(o, d, 6, accel)
            parity.check_hits(ref, hits, what)
            parity.check_radiance(ref_rad, rad, what)
        rgb, rgb8 = ctx.render(96, 72, mode=rt.MODE_REF_AA9, max_depth=6)
    finally:
        ctx.close()
    parity.check_radiance(ref_rgb, rgb, what)
    parity.check_rgb8(ref_rgb8, rgb8, what)


@pytest.mark.parametrize("n_lights,with_sphere", [(11, False), (9, True)])
def test_more_lights_than_the_scene_record_holds_on_gpu(n_lights, with_sphere):
    """Lights beyond the scene record's own table live right behind the record in device memory: a frame under 9 and
    11 lights (with and without the wave-wide skipping of shadow queries) against the oracle's, then again after
    rt_scene_update with the lights moved."""
    import random_scenes
    sp, what = random_scenes.many_lights(n_lights, with_sphere)
    ref_rgb, ref_rgb8, _ = sp.orc.render(96, 64, mode=rt.MODE_GRID, grid_m=2, max_depth=4)
    ctx = rt.Context(0)
    try:
        ctx.upload(sp.host)
        rgb, rgb8 = ctx.render(96, 64, mode=rt.MODE_GRID, grid_m=2, max_depth=4)
        ctx.update(sp.host)
        again, again8 = ctx.render(96, 64, mode=rt.MODE_GRID, grid_m=2, max_depth=4)
    finally:
        ctx.close()
    parity.check_radiance(ref_rgb, rgb, what)
    parity.check_rgb8(ref_rgb8, rgb8, what)
    assert np.array_equal(rgb, again, equal_nan=True) and np.array_equal(rgb8, again8)


@pytest.mark.parametrize("seed", range(3))
def test_sphere_ball_test_drops_no_hit_on_gpu(seed):
    """Spheres under hard transformations (axis ratios up to 60:1, tiny and far, far from the world's origin)
    and rays that graze them, leave them behind or start inside: the ball test in front of the exact sphere
    test never drops a hit the oracle, which has no such test, finds; shadow rays from a wall 4000 units
    away cover the shell of rounding "hits" that far-away rays see around a sphere."""
    import random_scenes
    sp, o, d = random_scenes.ellipsoids(seed)
    ref = sp.orc.find_closest_hit(o, d)
    ref_rad, _ = sp.orc.trace_rays(o, d, 6)
    ref_rgb, ref_rgb8, _ = sp.orc.render(96, 72, mode=rt.MODE_REF_AA9, max_depth=6)
    ctx = rt.Context(0)
    try:
        ctx.upload(sp.host)
        hits, rad = ctx.trace_rays(o, d, 6, rt.ACCEL_LBVH)
        rgb, rgb8 = ctx.render(96, 72, mode=rt.MODE_REF_AA9, max_depth=6)
    finally:
        ctx.close()
    parity.check_hits(ref, hits, "ellipsoids")
    parity.check_radiance(ref_rad, rad, "ellipsoids")
    parity.check_radiance(ref_rgb, rgb, "ellipsoids frame")
    parity.check_rgb8(ref_rgb8, rgb8, "ellipsoids frame")


def test_half_a_million_triangles():
    """A closed bumpy surface of 500 000 triangles behind a Box gate: upload, cut, sort, build (a tree of about
    twenty levels, one launch each) and fit stay within seconds, and the tree gives the records of the linear scan."""
    import time
    import random_scenes
    rng = np.random.default_rng(11)
    pos, nrm = random_scenes._shell(rng, 500, 500)
    assert len(pos) == 500000
    hs = rt.HostScene()
    hs.add_mesh_tris(pos, nrm, rt.scale(3.0, rt.translate((0, 0, 9))), rt.Material(diffuse=(0.7, 0.6, 0.3), specular=0.4, shininess=30.0),
                     boxed=True)
    hs.add_plane((0, -4, 0), (0, 1, 0), rt.Material(diffuse=(0.4, 0.4, 0.4)))
    hs.add_light((6, 9, 0), 1.0)
    ctx = rt.Context(0)
    try:
        t0 = time.perf_counter()
        ctx.upload(hs)
        dt = time.perf_counter() - t0
        c = ctx.counters()
        assert 15 <= c.bvh_max_depth <= 64 and dt < 20.0, (c.bvh_max_depth, dt)
        o, d, _ = O.camera_rays(64, 48)
        a, ra = ctx.trace_rays(o, d, 4, rt.ACCEL_LBVH)
        b, rb = ctx.trace_rays(o, d, 4, rt.ACCEL_BRUTE)
        assert (a["object"] == 0).sum() > 200
        assert a.tobytes() == b.tobytes() and np.array_equal(ra, rb, equal_nan=True)
        rgb, rgb8 = ctx.render(320, 180, mode=rt.MODE_GRID, grid_m=2)
        assert np.isfinite(rgb).all() and rgb8.max() > 50
    finally:
        ctx.close()


@pytest.mark.parametrize("threshold", ["0", "3000", "20000", "4000000000"])
def test_per_tile_kernel_choice_is_bit_identical(threshold, host_scene, monkeypatch):
    """k_render_hybrid renders a tile either with one block (a lane per pixel) or with four (taps spread over the
    waves), by the time the tile took in the frame before: whatever the threshold -- all tiles heavy from the
    second frame on, a mix, none -- every frame equals k_render's."""
    w, h = 200, 120
    monkeypatch.setenv("RT_RENDER_VARIANT", "0")
    ctx = rt.Context(0)
    ctx.upload(host_scene("teapot"))
    plain, plain8 = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=3)
    ctx.close()
    monkeypatch.setenv("RT_RENDER_VARIANT", "3")
    monkeypatch.setenv("RT_HYBRID_THRESHOLD", threshold)
    ctx = rt.Context(0)
    ctx.upload(host_scene("teapot"))
    try:
        for k in range(4):
            rgb, rgb8 = ctx.render(w, h, mode=rt.MODE_GRID, grid_m=3)
            assert ctx.counters().render_variant == 3
            assert np.array_equal(rgb8, plain8) and np.array_equal(rgb, plain, equal_nan=True), (threshold, k)
        # another frame size with the same context: the remembered times no longer fit, the pixels still do
        a, a8 = ctx.render(96, 64, mode=rt.MODE_GRID, grid_m=2)
        monkeypatch.setenv("RT_RENDER_VARIANT", "0")
    finally:
        ctx.close()
    ctx = rt.Context(0)
    ctx.upload(host_scene("teapot"))
    b, b8 = ctx.render(96, 64, mode=rt.MODE_GRID, grid_m=2)
    ctx.close()
    assert np.array_equal(a8, b8) and np.array_equal(a, b, equal_nan=True)


def test_command_line_driver_writes_the_same_frame(gpu_ctx, tmp_path):
    """rt_main, the counterpart of the reference's `./main out.ppm` (main.cpp:1212-1288): its ASCII PPM holds the
    frame the library renders through the Python binding, and a three-frame sequence differs from frame to frame."""
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "rt_main")
    if not os.path.exists(exe):
        pytest.fail("ray_tracer_amd/rt_main is missing: make -C ray_tracer_amd/csrc")
    out = tmp_path / "teapot.ppm"
    r = subprocess.run([exe, str(out), "--preset", "teapot", "--size", "96x64", "--mode", "grid2", "--assets", rt.ASSETS],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    tokens = out.read_text().split()
    assert tokens[:4] == ["P3", "96", "64", "255"]
    got = np.array(tokens[4:], np.int32).reshape(64, 96, 3)
    _, rgb8 = gpu_ctx("teapot").render(96, 64, mode=rt.MODE_GRID, grid_m=2)
    assert np.array_equal(got, rgb8)
    seq = tmp_path / "turn.ppm"
    r = subprocess.run([exe, str(seq), "--preset", "teapot", "--size", "64x48", "--frames", "3", "--p6", "--assets", rt.ASSETS],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    frames = [(tmp_path / f"turn_{k:04d}.ppm").read_bytes() for k in range(3)]
    assert all(f.startswith(b"P6") for f in frames) and len(set(frames)) == 3
    # several devices from one process (here the same GPU three times: a context and a host thread each): the same bytes
    multi = tmp_path / "multi.ppm"
    r = subprocess.run([exe, str(multi), "--preset", "teapot", "--size", "96x64", "--mode", "grid2", "--devices", "0,0,0", "--assets", rt.ASSETS],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert multi.read_bytes() == out.read_bytes() and "slowest of 3 devices" in r.stdout
    r = subprocess.run([exe, str(multi), "--preset", "teapot", "--size", "96x64", "--gpus", "64", "--assets", rt.ASSETS],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "error" in r.stderr  # a device that does not exist: a message, not a crash


def test_packet_walk_counters(gpu_ctx):
    """The counting kernel reports the packet walk's node visits once per wave and the rays that took part: between one
    ray and all 64 per visit, and as many lane visits as half the slab tests made inside the walk at most."""
    ctx = gpu_ctx("teapot")
    ctx.reset_counters()
    ctx.render(256, 192, want_rgb=False, mode=rt.MODE_GRID, grid_m=2, count_work=True)
    c = ctx.counters()
    assert c.node_steps > 0 and c.node_steps <= c.node_lane_visits <= 64 * c.node_steps
    assert 2 * c.node_lane_visits <= c.box_tests


@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_BRUTE])
def test_reference_head_frame_on_gpu(accel, gpu_ctx):
    """The one complete output the reference is known to produce: the frame main() writes at HEAD
    (main.cpp:1212-1288: the `shells` scene, 1024x768, 1 spp).  tests/golden/head_frame.npz holds that frame as the
    oracle renders it; its P3 file hashes to the sha256 recorded from the reference binary
    (tests/golden/make_head_frame.py refuses to write it otherwise, tests/test_oracle_anchors.py re-checks).  The GPU
    frame, with the tree and with the linear scan: +-1 LSB on >= 99.9 % of the pixels (in fact nearly all are equal),
    radiance within 1e-4 relative on the stored lattice."""
    g = np.load(os.path.join(GOLDEN, "head_frame.npz"))
    rgb, rgb8 = gpu_ctx("shells").render(1024, 768, mode=rt.MODE_REF_1SPP, max_depth=8, accel=accel)
    diff = parity.check_rgb8(g["rgb8"], rgb8, "HEAD frame")
    equal = float((diff == 0).mean())
    print(f"HEAD frame, accel {accel}: {int((diff == 0).sum())} of {diff.size} pixels equal, {int((diff > 1).sum())} beyond 1 LSB")
    assert equal >= 0.995, equal
    parity.check_radiance(g["rgb_lattice"], rgb[::4, ::4], "HEAD frame lattice")


def test_command_line_driver_writes_the_reference_head_frame(tmp_path):
    """`rt_main out.ppm` with no other argument is the reference's `./main out.ppm`: the HEAD scene at 1024x768, one
    sample per pixel, as an ASCII P3 file of the reference's byte layout (Image.cpp:8-23: header `P3\\nw h\\n255\\n`,
    `r g b` and two blanks per pixel, a newline per row); its pixels are within 1 LSB of the reference frame's."""
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "rt_main")
    if not os.path.exists(exe):
        pytest.fail("ray_tracer_amd/rt_main is missing: make -C ray_tracer_amd/csrc")
    out = tmp_path / "head.ppm"
    r = subprocess.run([exe, str(out), "--assets", rt.ASSETS], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    data = out.read_bytes()
    assert data.startswith(b"P3\n1024 768\n255\n")
    rows = data[len(b"P3\n1024 768\n255\n"):].split(b"\n")
    assert len(rows) == 769 and rows[-1] == b""
    assert all(row.endswith(b"  ") and row.count(b"  ") == 1024 for row in rows[:3] + rows[-4:-1])
    got = np.array(data.split()[4:], np.int32).reshape(768, 1024, 3)
    g = np.load(os.path.join(GOLDEN, "head_frame.npz"))
    # the same bytes as the oracle's writer produces for these pixels (byte-exact against the reference's Image.cpp elsewhere)
    ref_file = tmp_path / "oracle.ppm"
    O.write_ppm(str(ref_file), got.astype(np.uint8))
    assert ref_file.read_bytes() == data
    parity.check_rgb8(g["rgb8"], got, "rt_main HEAD frame")
