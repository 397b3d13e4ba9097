#!/usr/bin/env python3
"""Headline benchmark: teapot.obj scene, 3840x2160, 16 samples per pixel (4x4 grid).

    python bench.py --gpus N --steps K --warmup W

One step = one whole frame: every rank renders its interleaved 16x16 tiles with the HIP trace
kernel, the tile-major shards are all-gathered over RCCL (N > 1) and de-interleaved into the
row-major frame on every rank.  The frame is what the reference produces: 8-bit RGB after tone mapping
and quantisation (main.cpp:1184-1189, Image.cpp:35-39); --float-frame also delivers the linear float
frame (4x the bytes through the kernel's stores, the all-gather and the de-interleave).  Scene, LBVH and output buffers are resident in HBM before the
timed region.  Rank 0 prints one JSON line: metric = Mrays/s where one ray is one closest-hit
query that was executed (primary, shadow, reflection, refraction alike); the reference's own ray
count, which also has the shadow rays that were answered without a query, is printed beside it
(rays.answered_*).  Plus the roofline of the trace kernel and a CPU baseline (the oracle
restatement of the reference's brute-force loop, timed on a sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FP32 operations per unit of work (FMA = 2), used for roofline.achieved
FLOPS = {"tri_tests": 71, "box_tests": 24, "gate_tri_tests": 71, "quadric_tests": 95,
         "plane_tests": 20, "shade_evals": 70}  # gate_tri_tests: the Box gate's triangle tests that were entered
PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (256 CU x 2.4 GHz x 256 flop/clk)

WORKLOADS = {
    # name: (preset, width, height, mode, grid_m, max_depth)
    "teapot_4k_16spp": ("teapot", 3840, 2160, 2, 4, 8),
    "analytic_4k_aa9": ("analytic", 3840, 2160, 1, 1, 8),
    "shells_4k_64spp": ("shells", 3840, 2160, 2, 8, 8),
    "textured_8k_256spp": ("textured", 8192, 8192, 2, 16, 8),
    "teapot_small": ("teapot", 512, 288, 2, 2, 8),
}


def cpu_baseline(preset, width, height, mode, grid_m, max_depth):
    """Times the CPU oracle (the restatement of the reference's single-threaded brute-force loop) on bounded samples
    of the same frame: 8x8-pixel blocks on a regular lattice, all taps.  Once with OpenMP over the allowed cores
    (blocks every 24 pixels, 1/9 of the frame) and once on one thread, as the reference runs (blocks every 96
    pixels, 1/144 of the frame)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    L = O.lib()
    scene = O.Scene(preset)

    def sample(step, threads):
        L.orc_set_threads(threads)
        os_, ds_ = [], []
        blk = 8
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
        return total / dt / 1e6, f"8x8-pixel blocks every {step} px of the {width}x{height} frame ({len(o)} primary rays, {total} rays, {dt:.1f} s)"

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    many, many_sample = sample(24, threads)
    one, one_sample = sample(96, 1)
    return {"value": many, "unit": "Mrays/s", "cores": threads, "kind": "port", "sample": many_sample,
            "single_thread": {"value": one, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": one_sample}}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: starts the N rank processes itself
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) and returns their exit code.  Called
    before this process has imported torch or touched the GPU; the parent only waits."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def timed_steps(step, barrier, steps, warmup):
    """The timing contract: `warmup` untimed steps, then exactly `steps` steps between two barriers."""
    for _ in range(warmup):
        step(None)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    barrier()
    return time.perf_counter() - t0


def stub_main(args, world, rank):
    """The launcher, the rendezvous, the barriers and the max-over-ranks reduction of this script with a
    step that needs no GPU (gloo): what tests/test_bench_launcher.py runs on a machine without one."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    acc = torch.zeros(1, dtype=torch.float64)

    def step(_k):
        acc.add_(1.0)
        time.sleep(0.002 * (rank + 1))  # ranks of unequal speed: the slowest decides

    def barrier():
        if world > 1:
            dist.barrier()

    elapsed = timed_steps(step, barrier, args.steps, args.warmup)
    t = torch.tensor([elapsed], dtype=torch.float64)
    units = torch.tensor([float(acc) - args.warmup], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, t.clone())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    else:
        per_rank = [t.clone()]
    if rank == 0:
        print(json.dumps({"metric": "stub steps/s", "value": float(units) / float(t), "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "synthetic", "config": {"workload": "stub"},
                          "elapsed_per_rank_s": [float(x) for x in per_rank]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="teapot_4k_16spp", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", default="lbvh", choices=["lbvh", "brute"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--float-frame", action="store_true", help="also produce the linear float frame")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)  # launcher test without a GPU
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not started by a launcher: start the ranks, before anything here touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.stub_step:
        return stub_main(args, world, rank)

    import torch
    import torch.distributed as dist
    import ray_tracer_amd as rt

    n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU runtime)
    if n_dev < world or local_rank >= n_dev:
        raise SystemExit(f"bench.py --gpus {world}: rank {rank} needs GPU {local_rank}, but this node has {n_dev} GPU(s); "
                         "one GPU per rank is required")
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

    evs = []

    def step(k):
        if k is None:
            pipe.submit(p)
        else:
            evs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            pipe.submit(p, evs[-1])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if not evs:  # between warm-up and the timed steps
            ctx.reset_counters()
            torch.cuda.synchronize()

    elapsed = timed_steps(step, barrier, args.steps, args.warmup)
    c = ctx.counters()
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(args.steps, 1)

    # one extra frame with the counting kernel variant (outside the timed region) for the roofline
    ctx.reset_counters()
    pc = ctx.params(width, height, mode=mode, grid_m=grid_m, max_depth=max_depth, accel=accel,
                    rank=rank, world_size=world, count_work=True)
    pipe.submit(pc)
    torch.cuda.synchronize()
    cw = ctx.counters()
    count_names = ("tri_tests", "box_tests", "gate_tri_tests", "quadric_tests", "plane_tests", "shade_evals")
    flops_local = sum(getattr(cw, n) * FLOPS[n] for n in count_names)

    # every rank's figures on rank 0: [elapsed, kernel ms, rays, shadow queries run, flops of a launch, counts...]
    mine = torch.tensor([elapsed, kernel_ms, float(c.rays), float(c.shadow_queries), float(flops_local)] +
                        [float(getattr(cw, n)) for n in count_names] +
                        [float(cw.rays), float(cw.gate_tests), float(c.rays - c.rays_shadow + c.shadow_queries),
                         float(cw.node_steps), float(cw.node_lane_visits)],
                        dtype=torch.float64, device=dev)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    rows = torch.stack(rows).cpu().numpy()

    if rank == 0:
        elapsed = float(rows[:, 0].max())  # the slowest rank decides
        rays_total, shadow_queries = float(rows[:, 2].sum()), float(rows[:, 3].sum())
        ms_per_step = elapsed / args.steps * 1e3
        executed_total = float(rows[:, 7 + len(count_names)].sum())
        value = executed_total / elapsed / 1e6
        # roofline of the render kernel: algorithmic flops of each rank's launch over that launch's duration, per GPU
        per_rank_tflops = [float(r[4]) / (float(r[1]) * 1e-3) / 1e12 if r[1] > 0 else 0.0 for r in rows]
        achieved_tflops = sum(per_rank_tflops) / world
        counts = {n: int(rows[:, 5 + i].sum()) for i, n in enumerate(count_names)}
        counts["rays"] = int(rows[:, 5 + len(count_names)].sum())
        counts["gates"] = int(rows[:, 6 + len(count_names)].sum())
        # the packet walk: node visits per wave of 64 rays, and how many of the 64 rays took part in a visit on average
        counts["node_steps_waves"] = int(rows[:, 8 + len(count_names)].sum())
        node_lane_visits = float(rows[:, 9 + len(count_names)].sum())
        kernel_ms_max = float(rows[:, 1].max())
        traffic = pmc_flops = traffic_source = fetch_bytes = None
        pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        kernel_names = ("k_render", "k_render_split<1>", "k_render_split<2>", "k_render_hybrid", "k_render_queue")
        # the counter passes were made on the single-GPU launch: no figure for a rank's share
        if world == 1 and os.path.exists(pmc_path):
            try:
                rec = json.load(open(pmc_path)).get(args.workload)
                if isinstance(rec, dict) and rec.get("kernel") == kernel_names[c.render_variant]:
                    traffic, pmc_flops, fetch_bytes = rec.get("hbm_bytes"), rec.get("fp32_flops"), rec.get("fetch_bytes")
                    # (recorded rocprofv3 --pmc passes of this workload and kernel, not counters read during this run)
                    traffic_source = "profiles/hbm_traffic.json <- profiles/" + str(rec.get("source"))
            except Exception:
                traffic = pmc_flops = traffic_source = fetch_bytes = None
        out = {
            "metric": ("Mrays/s (teapot.obj, 3840x2160, 16 spp)" if args.workload == "teapot_4k_16spp" else f"Mrays/s ({args.workload})") +
                      "; a ray = an executed closest-hit query",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # one ray = one find_closest_hit of the reference (main.cpp:653, 718); a shadow ray whose query the kernel
            # proved unnecessary (nothing can lie between the point and the light) is counted but was not traced
            # `value` counts executed queries only; answered_* also count the shadow rays decided without one
            "rays": {"executed_per_frame": executed_total / args.steps, "executed_Mrays_per_s": value,
                     "answered_per_frame": rays_total / args.steps, "answered_Mrays_per_s": rays_total / elapsed / 1e6,
                     "shadow_queries_executed_per_frame": shadow_queries / args.steps},
            "config": {"workload": args.workload, "preset": preset, "width": width, "height": height,
                       "samples_per_pixel": 1 if mode == 0 else 9 if mode == 1 else grid_m * grid_m,
                       "max_depth": max_depth, "accel": args.accel, "tiles": total_tiles,
                       "frame": "rgb8+f32" if ff else "rgb8",
                       "parallelism": f"tiles{world}", "rays_per_frame": executed_total / args.steps,
                       "bvh_height": c.bvh_max_depth, "lbvh_build_ms": c.last_build_ms,
                       "tree_build": "radix" if os.environ.get("RT_BUILD") == "radix" else "binned surface-area splits",
                       "render_kernel": kernel_names[c.render_variant]},
            "roofline": {"bound": "valu", "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_TFLOPS, "traffic": traffic, "pmc_flops": pmc_flops,
                         "traffic_source": traffic_source, "pmc_source": traffic_source,
                         # tree, triangle and texture fetches through the L2 (TCC) against the HBM peak
                         "fetch_GBps": (fetch_bytes / (kernel_ms_max * 1e-3) / 1e9) if fetch_bytes and kernel_ms_max > 0 else None,
                         "hbm_GBps": (traffic / (kernel_ms_max * 1e-3) / 1e9) if traffic and kernel_ms_max > 0 else None,
                         "hbm_peak_GBps": 8000.0,
                         "kernel": kernel_names[c.render_variant], "kernel_ms": kernel_ms_max,
                         "kernel_ms_per_rank": [float(x) for x in rows[:, 1]],
                         "achieved_per_rank": per_rank_tflops,
                         "flops_per_unit": FLOPS, "counts_per_launch": counts,
                         "packet_fill": (node_lane_visits / (64.0 * counts["node_steps_waves"])) if counts["node_steps_waves"] else None,
                         "tri_tests_per_s": (counts["tri_tests"] + counts["gate_tri_tests"]) / (kernel_ms_max * 1e-3) if kernel_ms_max > 0 else 0.0},
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
