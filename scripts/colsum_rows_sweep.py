#!/usr/bin/env python3
"""Rows-per-block sweep of cusrl::colsum_chunked_kernel<true> (ReLU backward + bias-gradient partials) at the config-2
minibatch, [24 576, 256] and [24 576, 128]: one process per value of CUSRL_COLSUM_ROWS (the library reads it once)."""
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
    out = []
    for H in (256, 128):
        g, y = f(B, H), torch.relu(f(B, H))
        out.append(timeit(lambda: ops.relu_backward_bias(g, y, defer=True), 400))
        out.append(timeit(lambda: ops.relu_backward_bias(g, y), 400))
    print(f"rows/block {os.environ.get('CUSRL_COLSUM_ROWS', 'default'):>8}: [B,256] deferred {out[0]:6.2f} us, with finalize {out[1]:6.2f} us | "
          f"[B,128] deferred {out[2]:6.2f} us, with finalize {out[3]:6.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rows in sys.argv[1:] or ["16", "24", "32", "48", "64", "96", "128"]:
            subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "CUSRL_COLSUM_ROWS": rows}, check=True)
