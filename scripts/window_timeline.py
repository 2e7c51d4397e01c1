#!/usr/bin/env python3
"""All-queue timeline of the dispatches between the nth launch of kernel A and the first launch of kernel B behind it, from a
rocprofv3 kernel trace:

    python scripts/window_timeline.py <kernel_trace.csv> <A substring> <B substring> [--nth -2] [--before 3]

(e.g. ``count_flags_kernel`` .. ``ppo_loss_rowgroup``: everything between the start of pre_update and the first minibatch step)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "GEMM " + name[:14] + (" MT" + m.group(1) if m else "")
    return name[:100]


def main():
    path, first, last = sys.argv[1], sys.argv[2], sys.argv[3]
    nth = int(sys.argv[sys.argv.index("--nth") + 1]) if "--nth" in sys.argv else -2
    before = int(sys.argv[sys.argv.index("--before") + 1]) if "--before" in sys.argv else 3
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    anchors = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    a = anchors[nth]
    b = next((i for i in range(a + 1, len(rows)) if last in rows[i]["Kernel_Name"]), len(rows) - 1)
    origin = int(rows[a]["Start_Timestamp"])
    queues, ends = {}, {}
    busy_until = None
    for r in rows[max(a - before, 0) : b + 1]:
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = queues.setdefault(r["Queue_Id"], len(queues))
        idle = "" if busy_until is None or start <= busy_until else f"device idle {(start - busy_until) / 1e3:6.1f}"
        busy_until = end if busy_until is None else max(busy_until, end)
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        print(f"  t={(start - origin) / 1e3:8.1f}  {(end - start) / 1e3:7.2f} us  q{q}  grid {grid:>8s}  {short(r['Kernel_Name']):60s} {idle}")


if __name__ == "__main__":
    main()
