#!/usr/bin/env python3
"""Headline benchmark: teapot.obj scene, 3840x2160, 16 samples per pixel (4x4 grid).

    python bench.py --gpus N --steps K --warmup W

One step = one whole frame: every rank renders its interleaved 16x16 tiles with the HIP trace
kernel, the tile-major shards are all-gathered over RCCL (N > 1) and de-interleaved into the
row-major frame on every rank.  The frame is what the reference produces: 8-bit RGB after tone mapping
and quantisation (main.cpp:1184-1189, Image.cpp:35-39); --float-frame also delivers the linear float
frame (4x the bytes through the kernel's stores, the all-gather and the de-interleave).  Scene, LBVH and output buffers are resident in HBM before the
timed region.  Rank 0 prints one JSON line: metric = Mrays/s where one ray is one closest-hit
query (primary, shadow, reflection, refraction alike), plus the roofline of the trace kernel and
a CPU baseline (the oracle restatement of the reference's brute-force loop, timed on a sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FP32 operations per unit of work (FMA = 2), used for roofline.achieved
FLOPS = {"tri_tests": 71, "box_tests": 24, "gate_tests": 12 * 71, "quadric_tests": 95,
         "plane_tests": 20, "shade_evals": 70, "tri_final": 60}
PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (256 CU x 2.4 GHz x 256 flop/clk)

WORKLOADS = {
    # name: (preset, width, height, mode, grid_m, max_depth)
    "teapot_4k_16spp": ("teapot", 3840, 2160, 2, 4, 8),
    "analytic_4k_aa9": ("analytic", 3840, 2160, 1, 1, 8),
    "shells_4k_64spp": ("shells", 3840, 2160, 2, 8, 8),
    "textured_8k_256spp": ("textured", 8192, 8192, 2, 16, 8),
    "teapot_small": ("teapot", 512, 288, 2, 2, 8),
}


def cpu_baseline(preset, width, height, mode, grid_m, max_depth, budget_blocks=None):
    """Times the CPU oracle on a bounded sample of the same frame: 8x8-pixel blocks on a regular
    24-pixel lattice (1/9 of the pixels, all taps), traced with OpenMP over all allowed cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    L = O.lib()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    L.orc_set_threads(threads)
    scene = O.Scene(preset)
    os_, ds_ = [], []
    step, blk = 24, 8
    for y0 in range(0, height - blk + 1, step):
        for x0 in range(0, width - blk + 1, step):
            o, d, _ = O.camera_rays(width, height, 90.0, mode, grid_m, rect=(x0, y0, x0 + blk, y0 + blk))
            os_.append(o)
            ds_.append(d)
    o, d = np.concatenate(os_), np.concatenate(ds_)
    t0 = time.perf_counter()
    _, rays = scene.trace_rays(o, d, max_depth)
    dt = time.perf_counter() - t0
    total = int(rays.sum())
    return {"value": total / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"8x8-pixel blocks every 24 px of the {width}x{height} frame "
                      f"({len(o)} primary rays, {total} rays, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="teapot_4k_16spp", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", default="lbvh", choices=["lbvh", "brute"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--float-frame", action="store_true", help="also produce the linear float frame")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ray_tracer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    preset, width, height, mode, grid_m, max_depth = WORKLOADS[args.workload]
    accel = rt.ACCEL_LBVH if args.accel == "lbvh" else rt.ACCEL_BRUTE

    ctx = rt.Context(local_rank)
    scene = rt.HostScene(preset)
    ctx.upload(scene)

    # render on one stream, exchange + untile on another, two buffers each: frame k's all-gather overlaps
    # frame k+1's rendering (ray_tracer_amd/frames.py)
    from ray_tracer_amd.frames import FramePipeline
    ff = args.float_frame
    pipe = FramePipeline(ctx, width, height, rank=rank, world=world, device=local_rank, float_frame=ff)
    torch.cuda.set_stream(pipe.render_stream)
    p = ctx.params(width, height, mode=mode, grid_m=grid_m, max_depth=max_depth, accel=accel,
                   rank=rank, world_size=world)
    total_tiles = pipe.total_tiles
    dev = pipe.dev

    def step(events=None):
        pipe.submit(p, events)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.reset_counters()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    c = ctx.counters()
    rays_local = c.rays
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(args.steps, 1)

    tl = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    tr = torch.tensor([float(rays_local)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
    elapsed, kernel_ms = float(tl[0]), float(tl[1])
    rays_total = float(tr[0])

    # one extra frame with the counting kernel variant (outside the timed region) for the roofline
    ctx.reset_counters()
    pc = ctx.params(width, height, mode=mode, grid_m=grid_m, max_depth=max_depth, accel=accel,
                    rank=rank, world_size=world, count_work=True)
    pipe.submit(pc)
    torch.cuda.synchronize()
    cw = ctx.counters()
    flops_local = (cw.tri_tests * FLOPS["tri_tests"] + cw.box_tests * FLOPS["box_tests"] +
                   cw.gate_tests * FLOPS["gate_tests"] + cw.quadric_tests * FLOPS["quadric_tests"] +
                   cw.plane_tests * FLOPS["plane_tests"] + cw.shade_evals * FLOPS["shade_evals"])
    achieved_tflops = flops_local / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        kernel_names = ("k_render", "k_render_split<1>", "k_render_split<2>", "k_render_hybrid")
        # the counter passes were made on the single-GPU launch of k_render: no figure for a rank's share
        if world == 1 and c.render_variant == 0 and os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (teapot.obj, 3840x2160, 16 spp)" if args.workload == "teapot_4k_16spp" else f"Mrays/s ({args.workload})",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "preset": preset, "width": width, "height": height,
                       "samples_per_pixel": 1 if mode == 0 else 9 if mode == 1 else grid_m * grid_m,
                       "max_depth": max_depth, "accel": args.accel, "tiles": total_tiles,
                       "frame": "rgb8+f32" if ff else "rgb8",
                       "parallelism": f"tiles{world}", "rays_per_frame": rays_total / args.steps,
                       "bvh_height": c.bvh_max_depth, "lbvh_build_ms": c.last_build_ms,
                       "tree_build": "radix" if os.environ.get("RT_BUILD") == "radix" else "binned surface-area splits",
                       "render_kernel": kernel_names[c.render_variant]},
            "roofline": {"bound": "valu", "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "kernel": kernel_names[c.render_variant], "kernel_ms": kernel_ms,
                         "counts_per_launch": {"tri_tests": cw.tri_tests, "box_tests": cw.box_tests,
                                               "gate_tests": cw.gate_tests, "quadric_tests": cw.quadric_tests,
                                               "plane_tests": cw.plane_tests, "shade_evals": cw.shade_evals,
                                               "rays": cw.rays},
                         "tri_tests_per_s": cw.tri_tests / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(preset, width, height, mode, grid_m, max_depth)
        print(json.dumps(out), flush=True)
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
