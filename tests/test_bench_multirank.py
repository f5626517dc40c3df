"""bench.py's multi-rank control flow, end to end on CPU: `--selftest-emu` runs the script's own rank set-up, per-rank shards, the
three-phase gradient exchange (parallel.GradientExchange over gloo), the barrier + max-over-ranks timing and the JSON line with the
host-emulated kernels at a tiny size.  What is checked is the CONTRACT (n_gpus, exchange, value = global batch x steps / time), not a
rate.  Also: `--gpus N` must agree with the torch.distributed environment, and without one bench.py launches the ranks itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _env():
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"] and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True
    ex = d["exchange"]
    assert ex is not None and ex["world_size"] == 2 and ex["backend"] == "gloo" and ex["payload_MB"] > 0 and len(ex["ranges"]) == 3
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert "selftest" in d and d["cpu_baseline"] is None


def test_two_ranks_under_torchrun():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--selftest-emu"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_json_line(r.stdout))


def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the script launches one rank per GPU itself instead of silently running one."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest-emu"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_json_line(r.stdout))


def test_gpus_flag_must_match_world_size():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--selftest-emu"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
