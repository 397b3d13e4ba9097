"""The N > 1 path on CPU: two processes (gloo) each produce their rank's tile shard, all-gather the
equal-sized shards and reassemble the frame with the same tile rule the GPU kernels use.  The shard
contents come from the CPU twin (test scaffolding), so this checks the sharding / gather /
de-interleave arithmetic and the process-group plumbing, not the kernels."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import ray_tracer_amd as rt
    from ray_tracer_amd import tiling
    import twin_lib as T

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tw = T.Twin(rt.HostScene("analytic"))
    full, _, _ = tw.render(w, h, mode=rt.MODE_REF_1SPP)  # each rank could render only its tiles; cut them here
    shard_np = tiling.shard_from_frame(full, rank, world)
    shard = torch.from_numpy(shard_np.copy()).flatten()  # flat, like the device buffers bench.py gathers
    gathered = torch.zeros(world * shard.numel(), dtype=shard.dtype)
    dist.all_gather_into_tensor(gathered, shard)
    frame = tiling.frame_from_shards(gathered.numpy().reshape((world,) + shard_np.shape), w, h)
    np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame)
    np.save(os.path.join(out_dir, f"full{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(80, 48), (50, 37)])
def test_two_rank_gather_reassembles_frame(size, tmp_path):
    import torch.multiprocessing as mp
    w, h = size
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, w, h, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / "frame0.npy"), np.load(tmp_path / "frame1.npy")
    full = np.load(tmp_path / "full0.npy")
    assert np.array_equal(f0, f1) and np.array_equal(f0, full)
