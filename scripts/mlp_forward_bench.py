#!/usr/bin/env python3
"""The one-launch inference pass (cusrl_mlp2_forward) next to the library chain it replaces, graph-timed (20 launches per
replay), at the sizes the loop runs it: acting (4096 / 8192 rows, with the sampling epilogue), the value pass and the statistics
pass over the whole buffer (98 304 rows).

    python scripts/mlp_forward_bench.py
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402


def graph_time(fn, launches=20, replays=20):
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(launches):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / replays / launches * 1e6


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    K, H1, H2 = 48, 256, 128
    for rows, A, sample in ((4096, 12, True), (8192, 12, True), (4096, 1, False), (98304, 1, False), (98304, 12, False), (1048576, 12, False)):
        w1, b1 = torch.randn(H1, K, device=dev) * 0.1, torch.randn(H1, device=dev) * 0.1
        w2, b2 = torch.randn(H2, H1, device=dev) * 0.1, torch.randn(H2, device=dev) * 0.1
        w3, b3 = torch.randn(A, H2, device=dev) * 0.1, torch.randn(A, device=dev) * 0.1
        x = torch.randn(rows, K, device=dev)
        eps, std = torch.randn(rows, A, device=dev), torch.rand(A, device=dev) + 0.5
        layers = (w1, b1, w2, b2, w3, b3)

        def fused():
            return ops.mlp2_forward(x, layers, std=std, eps=eps) if sample else ops.mlp2_forward(x, layers)

        def library():
            h = torch._addmm_activation(b1, x, w1.t())
            h = torch._addmm_activation(b2, h, w2.t())
            if sample:
                return ops.normal_sample_logp(torch.mm(h, w3.t()), std, eps, repeat_std=True, mean_bias=b3)
            if A == 1:
                return ops.narrow_linear_forward(h, w3, b3)
            return torch.addmm(b3, h, w3.t())

        with torch.no_grad():
            f, l = graph_time(fused), graph_time(library)
        flops = 2.0 * rows * (K * H1 + H1 * H2 + H2 * A)
        print(f"rows {rows:8d}  out {A:2d}  {'acting (sampling epilogue)' if sample else 'plain':27s} fused {f:8.2f} us ({flops / f / 1e6:6.1f} TFLOP/s)"
              f"   library chain {l:8.2f} us")


if __name__ == "__main__":
    main()
