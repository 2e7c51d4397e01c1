#!/usr/bin/env python3
"""Kernel sequence between two consecutive launches of an anchor kernel, from a rocprofv3 kernel trace:

    python scripts/kernel_sequence.py <kernel_trace.csv> <anchor substring> [--nth -3]

prints the dispatches (short name, grid, duration, gap to the previous kernel's end) from the nth anchor to the next one —
e.g. one env step of the captured rollout (anchor `normal_sample_logp`) or one minibatch step (anchor `gather_kernel`)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "rocblas/hipblaslt GEMM " + name[:14] + (" MT" + m.group(1) if m else "")
    return name[:110]


def main():
    path, anchor = sys.argv[1], sys.argv[2]
    nth = int(sys.argv[sys.argv.index("--nth") + 1]) if "--nth" in sys.argv else -3
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    anchors = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(anchors) < 2:
        print(f"fewer than two launches of '{anchor}'")
        return
    a = anchors[nth]
    later = [i for i in anchors if i > a]
    b = later[0] if later else len(rows)
    prev_end = int(rows[a - 1]["End_Timestamp"]) if a else int(rows[a]["Start_Timestamp"])
    total = int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    print(f"{b - a} dispatches, {total / 1e3:.1f} us from the start of the first to the end of the last")
    for r in rows[a:b]:
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        print(f"  {(end - start) / 1e3:7.2f} us  gap {(start - prev_end) / 1e3:6.2f}  grid {grid:>9s}  {short(r['Kernel_Name'])}")
        prev_end = end


if __name__ == "__main__":
    main()
