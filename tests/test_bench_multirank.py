"""bench.py's multi-rank control flow, end to end on CPU: tests/bench_on_emulator.py calls bench.main with the host-emulated kernel
build injected (bench.py itself has no test hook and imports nothing from tests/), which runs the script's own rank set-up, per-rank
shards, the three-phase gradient exchange (parallel.GradientExchange over gloo), the barrier + max-over-ranks timing and the JSON
line at a tiny size.  What is checked is the CONTRACT (n_gpus, exchange, value = global batch x steps / time), not a
rate.  Also: `--gpus N` must agree with the torch.distributed environment, and without one bench.py launches the ranks itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(REPO, "tests", "bench_on_emulator.py")


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
    assert ex["collective"] == "torch.distributed.all_reduce (gloo)"
    pr = d["per_rank"]                                   # every rank's own time and rate: a straggler is visible in the line
    assert len(pr["ms_per_step"]) == 2 and len(pr["value"]) == 2 and pr["slowest_rank"] in (0, 1)
    assert abs(max(pr["ms_per_step"]) - d["ms_per_step"]) < 1e-3 and pr["ms_per_step_min"] <= pr["ms_per_step_max"]


def test_bench_script_has_no_test_hook():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "selftest-emu" not in src and "from tests" not in src and "import tests" not in src


def test_two_ranks_under_torchrun():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), DRIVER, "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_json_line(r.stdout))


def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the script launches one rank per GPU itself instead of silently running one."""
    cmd = [sys.executable, DRIVER, "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_json_line(r.stdout))


def test_gpus_flag_must_match_world_size():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, DRIVER, "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
