#!/usr/bin/env python3
"""Rows-per-block sweep of cusrl::narrow_linear_bwd_kernel at the config-2 minibatch (B = 24 576, K = 128): one process per
value of CUSRL_HEAD_ROWS (the library reads it once).  `python scripts/narrow_head_sweep.py` prints one line per value."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def child():
    sys.path[:0] = [str(ROOT), str(ROOT / "scripts")]
    import torch

    from cusrl_amd import ops
    from kernel_bench import timeit

    B = 24576
    f = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    h = torch.relu(f(B, 128))
    out = []
    for O in (12, 1):
        g, w = f(B, O), f(O, 128)
        out.append(timeit(lambda: ops.narrow_linear_backward(g, h, w, relu_input=True, defer=True), 400))
        out.append(timeit(lambda: ops.narrow_linear_backward(g, h, w), 400))
    print(f"rows/block {os.environ.get('CUSRL_HEAD_ROWS', 'default'):>8}: 128->12 relu+deferred {out[0]:6.2f} us, plain {out[1]:6.2f} us | "
          f"128->1 relu+deferred {out[2]:6.2f} us, plain {out[3]:6.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rows in sys.argv[1:] or ["24", "32", "48", "64", "96", "128", "192"]:
            subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "CUSRL_HEAD_ROWS": rows}, check=True)
