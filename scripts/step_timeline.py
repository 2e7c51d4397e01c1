#!/usr/bin/env python3
"""Timeline of consecutive minibatch steps from a rocprofv3 kernel trace (one line per dispatch, all queues interleaved):

    python scripts/step_timeline.py <kernel_trace.csv> <anchor substring> [--nth -6] [--steps 2]

start relative to the first anchor, duration, the hardware queue the dispatch ran on and the gap to the previous dispatch's
end ON THAT QUEUE — what shows whether a step's branches overlap, where the forks and joins sit and whether the gather runs
under the previous step.  Also prints the anchor-to-anchor period over every pair of consecutive anchors of the trace that
lie in the same update (periods < 1 ms)."""
import csv
import re
import statistics
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "GEMM " + name[:14] + (" MT" + m.group(1) if m else "")
    return name[:90]


def main():
    path, anchor = sys.argv[1], sys.argv[2]
    nth = int(sys.argv[sys.argv.index("--nth") + 1]) if "--nth" in sys.argv else -6
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    anchors = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(anchors) < steps + 2:
        print(f"too few launches of '{anchor}'")
        return
    starts = [int(rows[i]["Start_Timestamp"]) for i in anchors]
    periods = [(b - a) / 1e3 for a, b in zip(starts, starts[1:]) if b - a < 1_000_000]
    if periods:
        print(f"anchor-to-anchor period over {len(periods)} in-update pairs: median {statistics.median(periods):.1f} us, "
              f"mean {statistics.mean(periods):.1f} us, min {min(periods):.1f}, max {max(periods):.1f}")
    if "--periods" in sys.argv:  # the periods of the last `count` anchor pairs in order (an update's steps: where do the long ones sit?)
        count = int(sys.argv[sys.argv.index("--periods") + 1])
        tail = starts[-(count + 1):]
        print("last periods (us): " + " ".join(f"{(b - a) / 1e3:.0f}" for a, b in zip(tail, tail[1:])))
    a = anchors[nth]
    b = anchors[nth + steps] if nth + steps < 0 or nth + steps < len(anchors) else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    queue_key = next((k for k in ("Queue_Id", "Stream_Id") if k in rows[0]), None)
    last_end: dict[str, int] = {}
    print(f"{b - a} dispatches over {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us ({steps} steps)")
    for r in rows[max(a - 3, 0):b]:
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get(queue_key, "?") if queue_key else "?"
        gap = (start - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = end
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        print(f"  t={((start - t0) / 1e3):8.2f}  {(end - start) / 1e3:7.2f} us  q{q:<3s} gap {gap:7.2f}  grid {grid:>8s}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
