#!/usr/bin/env python3
"""Stand-alone timing of the first-layer backward: cusrl_input_layer_bwd next to what it replaces (cusrl_relu_bwd_colsum + the
split-batch weight-gradient GEMM), graph-timed like bench.py's kernels.    python scripts/input_layer_bench.py [rows K H]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import graph_time  # noqa: E402
from cusrl_amd import ops  # noqa: E402


def main():
    rows, K, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (24576, 48, 256)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = torch.randn(rows, H, device=dev)
    y = torch.relu(torch.randn(rows, H, device=dev))
    x = torch.randn(rows, K, device=dev)
    splits = 16

    def old():
        masked, _ = ops.relu_backward_bias(g, y, defer=True)
        return torch.bmm(masked.view(splits, rows // splits, H).transpose(1, 2), x.view(splits, rows // splits, K))

    def colsum_only():
        return ops.relu_backward_bias(g, y, defer=True)

    def new():
        return ops.input_layer_backward(g, y, x)

    nbytes_new = rows * 4 * (2 * H + K)
    for name, fn, nbytes in (("relu_bwd_colsum + bmm (rounds 2-5)", old, rows * 4 * (3 * H) + rows * 4 * (H + K)),
                             ("relu_bwd_colsum alone", colsum_only, rows * 4 * 3 * H),
                             ("cusrl_input_layer_bwd", new, nbytes_new)):
        us = graph_time(fn, launches=10, replays=20)
        print(f"{name:40s} {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s algorithmic ({nbytes / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
