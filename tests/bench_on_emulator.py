"""TEST DRIVER (tests/test_bench_multirank.py), never a measurement: calls bench.main with the host-emulated kernel build injected,
so that bench.py's own control flow - the --gpus / WORLD_SIZE check, self-spawning under torch.distributed.run, per-rank shards, the
three-phase gradient exchange over gloo, barrier + max-over-ranks timing, the JSON line - runs end to end on CPU at a tiny size
(16x16 fields, Unet3D dim=8).  bench.py itself imports nothing from tests/."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from tests.emu_util import emu_lib  # noqa: E402

if __name__ == "__main__":
    bench.main(test_env={"lib": emu_lib(), "image": 16, "dim": 8, "batch": 2})
