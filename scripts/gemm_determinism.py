#!/usr/bin/env python3
"""Are the library GEMMs of a minibatch step bit-reproducible call to call?  (rocBLAS may pick split-K kernels that
accumulate with atomics.)  Each shape: 300 calls on the same operands, eager and replayed from a hipGraph."""
import os
import sys

import torch

if os.environ.get("DEBUG_DETERMINISTIC") == "1":
    torch.use_deterministic_algorithms(True, warn_only=True)
dev = "cuda:0"
torch.manual_seed(0)


def check(name, fn, calls=300):
    ref = fn().clone()
    eager = sum(int(not torch.equal(fn(), ref)) for _ in range(calls))
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        out = fn()
    replayed = 0
    for _ in range(calls):
        graph.replay()
        torch.cuda.synchronize()
        replayed += int(not torch.equal(out, ref))
    print(f"{name:58s} eager mismatches {eager:3d}/{calls}  replay mismatches {replayed:3d}/{calls}", flush=True)


for B in (1024, 24576):
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    x48, x12, h256, h128 = f(B, 48), f(B, 12), f(B, 256), f(B, 128)
    w1, w2, wd = f(256, 48), f(128, 256), f(256, 12)
    g128, g256 = f(B, 128), f(B, 256)
    check(f"B={B} dX2  [B,128] @ [128,256]", lambda: g128 @ w2)
    check(f"B={B} dW2  [128,B] @ [B,256]", lambda: g128.t() @ h256)
    check(f"B={B} dW1  [256,B] @ [B,48]", lambda: g256.t() @ x48)
    check(f"B={B} dW1  [256,B] @ [B,12]", lambda: g256.t() @ x12)
    check(f"B={B} fwd1 addmm_relu [B,48] x [48,256]", lambda: torch._addmm_activation(torch.zeros(256, device=dev), x48, w1.t()))
    check(f"B={B} fwd1 addmm_relu [B,12] x [12,256]", lambda: torch._addmm_activation(torch.zeros(256, device=dev), x12, wd.t()))
    check(f"B={B} fwd2 addmm [B,256] x [256,128]", lambda: torch.addmm(torch.zeros(128, device=dev), h256, w2.t()))
