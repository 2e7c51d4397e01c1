#!/usr/bin/env python3
"""Graph-timed census of the GEMMs one PPO minibatch step issues (config 2: B = 24576, MLP 48-256-128, heads 12 / 1),
next to their HBM floor, plus candidate reshapes (actor and critic layers side by side in one launch).

    python scripts/gemm_census.py [--rows 24576]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from kernel_bench import timeit  # noqa: E402

DEV = "cuda:0"


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--rows", type=int, default=24576)
    parser.add_argument("--iters", type=int, default=200)
    args = parser.parse_args()
    B, S = args.rows, 16
    r = lambda *s: torch.randn(*s, device=DEV)  # noqa: E731
    x, h1, h2 = r(B, 48), r(B, 256), r(B, 128)
    w1, b1, w2, b2 = r(256, 48), r(256), r(128, 256), r(128)
    wm, bm, wv, bv = r(12, 128), r(12), r(1, 128), r(1)
    g1, g2, gm, gv = r(B, 256), r(B, 128), r(B, 12), r(B, 1)
    w1x2, b1x2 = r(512, 48), r(512)
    g1x2 = r(B, 512)
    h1x2 = r(B, 512)
    w2x2 = r(2, 128, 256)
    g2x2 = r(B, 256)
    rows = []

    def add(name, fn, nbytes):
        us = timeit(fn, args.iters)
        rows.append((name, us, nbytes / 1e6, nbytes / 5.0e6))  # floor at 5 TB/s achievable

    f4 = 4
    add("fwd L1  addmm+relu [B,48]x[48,256]", lambda: torch._addmm_activation(b1, x, w1.t()), f4 * (B * 48 + B * 256))
    add("fwd L2  addmm+relu [B,256]x[256,128]", lambda: torch._addmm_activation(b2, h1, w2.t()), f4 * (B * 256 + B * 128))
    add("fwd mean head [B,128]x[128,12]", lambda: torch.addmm(bm, h2, wm.t()), f4 * (B * 128 + B * 12))
    add("fwd value head [B,128]x[128,1]", lambda: torch.addmm(bv, h2, wv.t()), f4 * (B * 128 + B))
    add("bwd dX mean head [B,12]x[12,128]", lambda: gm @ wm, f4 * (B * 12 + B * 128))
    add("bwd dX value head [B,1]x[1,128]", lambda: gv @ wv, f4 * (B + B * 128))
    add("bwd dX L2 [B,128]x[128,256]", lambda: g2 @ w2, f4 * (B * 128 + B * 256))
    add("bwd dW L2 bmm16 + sum", lambda: torch.bmm(g2.view(S, B // S, 128).transpose(1, 2), h1.view(S, B // S, 256)).sum(0), f4 * (B * 128 + B * 256))
    add("bwd dW L1 bmm16 + sum", lambda: torch.bmm(g1.view(S, B // S, 256).transpose(1, 2), x.view(S, B // S, 48)).sum(0), f4 * (B * 256 + B * 48))
    add("bwd dW mean bmm16 + sum", lambda: torch.bmm(gm.view(S, B // S, 12).transpose(1, 2), h2.view(S, B // S, 128)).sum(0), f4 * (B * 12 + B * 128))
    add("bwd dW value bmm16 + sum", lambda: torch.bmm(gv.view(S, B // S, 1).transpose(1, 2), h2.view(S, B // S, 128)).sum(0), f4 * (B + B * 128))
    # ---- candidates: both towers in one launch
    add("CAND fwd L1 x2 (N concat) [B,48]x[48,512]", lambda: torch._addmm_activation(b1x2, x, w1x2.t()), f4 * (B * 48 + B * 512))
    add("CAND fwd L2 x2 (bmm 2, strided A)", lambda: torch.bmm(h1x2.view(B, 2, 256).transpose(0, 1), w2x2.transpose(1, 2)), f4 * 2 * (B * 256 + B * 128))
    add("CAND dW L1 x2 bmm16 + sum", lambda: torch.bmm(g1x2.view(S, B // S, 512).transpose(1, 2), x.view(S, B // S, 48)).sum(0), f4 * (B * 512 + B * 48))
    add("CAND dX L2 x2 (bmm 2)", lambda: torch.bmm(g2x2.view(B, 2, 128).transpose(0, 1), w2x2), f4 * 2 * (B * 128 + B * 256))
    add("CAND dW L2 x2 bmm32 + sum", lambda: torch.bmm(
        g2x2.view(S, B // S, 2, 128).permute(2, 0, 3, 1).reshape(2 * S, 128, B // S),
        h1x2.view(S, B // S, 2, 256).permute(2, 0, 1, 3).reshape(2 * S, B // S, 256)).view(2, S, 128, 256).sum(1), f4 * 2 * (B * 128 + B * 256))
    add("CAND heads x2 fwd [B,128]x[128,13] (one tower)", lambda: torch.addmm(r(13), h2, r(13, 128).t()), f4 * (B * 128 + B * 13))
    print(f"{'gemm':58s} {'us':>8s} {'MB':>8s} {'floor us':>9s} {'x floor':>8s}")
    for name, us, mb, floor in rows:
        print(f"{name:58s} {us:8.2f} {mb:8.2f} {floor:9.2f} {us / floor:8.2f}")


if __name__ == "__main__":
    main()
