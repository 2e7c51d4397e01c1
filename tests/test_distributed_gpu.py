"""The data-parallel path on RCCL (backend ``nccl``) with the single GPU a test box has: one torchrun rank."""

import json
import math
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("compile_", ["0", "1"])
def test_ppo_preset_over_rccl_single_rank(tmp_path, compile_):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_dist_gpu_worker.py"), str(out), compile_]
    done = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    result = json.loads(out.read_text())
    assert result["world"] == 1 and result["mean"] == 1.0 and result["var"] == 4.0
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/entropy_loss", "Agent/kl_divergence"):
        assert math.isfinite(result["info"][key]), key
