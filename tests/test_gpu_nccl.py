"""RCCL on hardware with one rank: `init_process_group("nccl")`, `parallel.run_sharded` (loss all-reduce
on the engine's stream next to graph replay, final all-gather) and two exchange families
(`HipSVI.run_exchanged`) equal the fits without a process group.  -m gpu.

Runs in a child process: a process group is process-global state, and a failing RCCL start-up must
not take the rest of the GPU suite with it."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_nccl_sharded_fits_equal_the_plain_fits():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_one_rank.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "NCCL_ONE_RANK_OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
