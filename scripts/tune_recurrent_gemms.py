#!/usr/bin/env python3
"""Measure the rocBLAS / hipBLASLt kernel selection for the per-time-step GEMMs of the fused GRU core (nn/gru.py):
``h[:rows] @ W_hh^T`` and ``dh[:rows] += d_gh[:rows] @ W_hh`` with ``rows`` a multiple of 256 (``_gemm_rows``).

    python scripts/tune_recurrent_gemms.py --out gpurun_out/tuned_recurrent.csv [--hidden 256] [--max-rows 8192]
    python scripts/merge_tuned_gemms.py gpurun_out/tuned_recurrent.csv
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["CUSRL_TUNED_GEMMS"] = "0"
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--hidden", type=int, nargs="+", default=[256])
    parser.add_argument("--max-rows", type=int, default=8192)
    parser.add_argument("--max-ms", type=int, default=20)
    parser.add_argument("--out", required=True)
    args = parser.parse_args()
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(args.max_ms)
    tunable.set_max_tuning_iterations(100)
    tunable.set_filename(args.out)
    device = "cuda:0"
    for H in args.hidden:
        B = args.max_rows
        w_hh = torch.randn(3 * H, H, device=device) * 0.05
        h = torch.randn(B, H, device=device)
        gh = torch.empty(B, 3 * H, device=device)
        dh = torch.zeros(B, H, device=device)
        for rows in range(256, B + 1, 256):
            torch.mm(h[:rows], w_hh.t(), out=gh[:rows])
            dh[:rows].addmm_(gh[:rows], w_hh)
        torch.cuda.synchronize()
        print(f"hidden {H}: {len(tunable.get_results())} entries", flush=True)
    for line in tunable.get_results()[:6]:
        print(line)


if __name__ == "__main__":
    main()
