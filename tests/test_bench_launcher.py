"""`python bench.py --gpus N` must start its own ranks when no launcher did (WORLD_SIZE unset), and it must
also run under torch.distributed.run.  Both paths are exercised here on CPU with bench.py's stub step
(gloo, no GPU): the launcher, the rendezvous on 127.0.0.1, the barriers around the timed steps and the
max-over-ranks reduction are the ones the GPU benchmark uses."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-step"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 1
    # ten steps in all; rank 1 sleeps twice as long per step, and the slowest rank decides the time
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 2.0) < 1e-6
    assert len(j["elapsed_per_rank_s"]) == 2 and j["ms_per_step"] >= 4.0


def test_bench_runs_under_the_drivers_launcher():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)["n_gpus"] == 2


def test_mismatched_world_is_an_error():
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2" in r.stderr


def test_single_rank_stub():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--stub-step"], env=_clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)["n_gpus"] == 1
