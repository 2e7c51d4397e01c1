#!/usr/bin/env python3
"""Where the device idles during one training iteration, from a rocprofv3 kernel trace:

    python scripts/idle_gaps.py <kernel_trace.csv> [--anchor <kernel name part>] [--min-us 12]

takes the LAST full iteration (from the first dispatch of the anchor kernel's last-but-one burst to the next burst), forms the
union of all queues' busy intervals and prints every idle gap longer than --min-us with the kernels on either side — host
read-backs, replay boundaries and cross-queue joins show up as such gaps — plus the busy / idle totals."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "GEMM " + name[:14] + (" MT" + m.group(1) if m else "")
    return name[:70]


def main():
    path = sys.argv[1]
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else None
    min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 12.0
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    # a kernel that only the rollout launches, once per env step (since round 6's second part acting is cusrl_mlp2_forward, which
    # the value and statistics passes launch too: the env's own step kernel / the step epilogue mark the rollout)
    for candidate in ([anchor] if anchor else ["normal_sample_logp", "synthetic_env_step", "step_epilogue"]):
        hits = [i for i, r in enumerate(rows) if candidate in r["Kernel_Name"]]
        if hits:
            break
    if not hits:
        print("no anchor kernel in the trace")
        return
    # bursts of the anchor (one per rollout): split where consecutive hits are > 2 ms apart
    bursts = [hits[0]]
    for a, b in zip(hits, hits[1:]):
        if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 2_000_000:
            bursts.append(b)
    if len(bursts) < 3:
        print("fewer than three rollouts in the trace")
        return
    first, last = bursts[-3], bursts[-2]
    span = rows[first:last]
    t0, t1 = int(span[0]["Start_Timestamp"]), int(rows[last]["Start_Timestamp"])
    print(f"iteration of {(t1 - t0) / 1e3:.1f} us, {len(span)} dispatches")
    busy_end, busy, gaps = t0, 0, []
    prev = None
    for r in span:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > busy_end:
            if s - busy_end >= min_us * 1e3:
                gaps.append((busy_end - t0, s - busy_end, prev, r))
            busy += e - s
            busy_end = e
        elif e > busy_end:
            busy += e - busy_end
            busy_end = e
        if prev is None or e >= int(prev["End_Timestamp"]):
            prev = r
    idle = (t1 - t0) - busy
    print(f"device busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us ({100 * idle / (t1 - t0):.1f} %); gaps >= {min_us} us: "
          f"{sum(g[1] for g in gaps) / 1e3:.1f} us in {len(gaps)}")
    for at, length, before, after in gaps:
        print(f"  t={at / 1e3:8.1f}  idle {length / 1e3:7.1f} us  after {short(before['Kernel_Name']) if before else '-':45s} before {short(after['Kernel_Name'])}")


if __name__ == "__main__":
    main()
