#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of scripts/pmc_cases.py into profiles/r02/pmc_summary.json.

    python scripts/pmc_summarize.py <dir with cases.json, FETCH_SIZE_counters.csv, WRITE_SIZE_counters.csv> <out.json>

Rows are matched to cases by kernel name + grid size.  FETCH_SIZE / WRITE_SIZE come in KB (rocprofv3 derived metrics);
the gfx950 correction of the read counter is CALIBRATED here instead of assumed: ``stream_16B`` moves a known 1 GiB each
way, the three ``random_rows_*`` cases read a known number of distinct rows far beyond the caches.
"""
import csv
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def per_launch(path, cases):
    """Average counter value of each case's launches: the rows of the case's kernel, in dispatch order, at the ordinals
    scripts/pmc_cases.py recorded — and of the expected grid size (several cases share a grid size)."""
    by_kernel = {}
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda row: int(row["Dispatch_Id"]))
    for row in rows:
        for kernel in ("gather_kernel", "push_kernel", "pack_rows_kernel", "ppo_loss_rowgroup_kernel"):
            if kernel in row["Kernel_Name"]:
                by_kernel.setdefault(kernel, []).append(row)
    sums = {}
    for name, case in cases.items():
        mine = by_kernel.get(case["kernel"], [])[case["first_ordinal"]: case["first_ordinal"] + case["ordinals"]]
        sums[name] = [float(row["Counter_Value"]) for row in mine if int(row["Grid_Size"]) == case["grid_threads"]]
    return {name: (sum(v) / len(v) if v else None) for name, v in sums.items()}, {name: len(v) for name, v in sums.items()}


def sq_counters(path, cases):
    """{case: {counter: average per launch}} + the ratios that say where a wave's time goes.  SQ_WAVE_CYCLES, SQ_WAIT_ANY
    and SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md): ratios between them are unit-free."""
    by_kernel = {}
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda row: int(row["Dispatch_Id"]))
    dispatches = {}  # dispatch id -> (kernel, grid, {counter: value})
    for row in rows:
        for kernel in ("gather_kernel", "push_kernel", "pack_rows_kernel", "ppo_loss_rowgroup_kernel"):
            if kernel in row["Kernel_Name"]:
                entry = dispatches.setdefault(int(row["Dispatch_Id"]), (kernel, int(row["Grid_Size"]), {}))
                entry[2][row["Counter_Name"]] = float(row["Counter_Value"])
    for dispatch in sorted(dispatches):
        kernel, grid, counters = dispatches[dispatch]
        by_kernel.setdefault(kernel, []).append((grid, counters))
    out = {}
    for name, case in cases.items():
        mine = by_kernel.get(case["kernel"], [])[case["first_ordinal"]: case["first_ordinal"] + case["ordinals"]]
        mine = [counters for grid, counters in mine if grid == case["grid_threads"]]
        if not mine:
            continue
        average = {key: sum(c.get(key, 0.0) for c in mine) / len(mine) for key in mine[0]}
        wave = average.get("SQ_WAVE_CYCLES") or 0.0
        if wave:
            average["valu_issue_share_of_wave_cycles"] = round(average.get("SQ_ACTIVE_INST_VALU", 0.0) / wave, 4)
            average["lds_issue_share_of_wave_cycles"] = round(average.get("SQ_ACTIVE_INST_LDS", 0.0) / wave, 4)
            average["waiting_share_of_wave_cycles"] = round(average.get("SQ_WAIT_ANY", 0.0) / wave, 4)
        if average.get("SQ_LDS_IDX_ACTIVE"):
            average["lds_bank_conflict_share_of_lds_cycles"] = round(average.get("SQ_LDS_BANK_CONFLICT", 0.0) / average["SQ_LDS_IDX_ACTIVE"], 4)
        out[name] = average
    return out


def main(directory, out_path):
    directory = Path(directory)
    cases = json.loads((directory / "cases.json").read_text())
    fetch_kb, fetch_n = per_launch(directory / "FETCH_SIZE_counters.csv", cases)
    write_kb, write_n = per_launch(directory / "WRITE_SIZE_counters.csv", cases)
    out = {"unit": "bytes per launch", "counter_unit": "KB (x1024)", "cases": {}}
    stream = fetch_kb.get("stream_16B")
    factor_stream = (1 << 30) / (stream * 1024) if stream else None
    for name, case in cases.items():
        fetch = None if fetch_kb[name] is None else fetch_kb[name] * 1024
        write = None if write_kb[name] is None else write_kb[name] * 1024
        out["cases"][name] = {"algorithmic_bytes": case["algorithmic_bytes"], "fetch_raw": fetch, "write_raw": write,
                              "rows_matched": [fetch_n[name], write_n[name]]}
    calibration = {"stream_16B_read_factor": factor_stream}
    for name, rows, row_bytes in (("random_rows_4B", 1 << 23, 4), ("random_rows_32B", 1 << 23, 32), ("random_rows_192B", 1 << 21, 192)):
        raw = out["cases"][name]["fetch_raw"]
        if raw:
            index_bytes = rows * 8  # the int64 index vector is a streaming read counted at the streaming factor
            row_raw = raw - index_bytes / (factor_stream or 2.0)
            calibration[name] = {"raw_fetch_bytes_per_row": row_raw / rows, "row_bytes": row_bytes}
    out["calibration"] = calibration
    # the read counter tallies each memory-side request at 64 B: a streaming 128-byte request counts half (factor 2);
    # a random row shorter than a request is ONE request (raw ~64 B per 4- or 32-byte row), so for the row part of a
    # gather raw x 2 is an UPPER bound (request = 128 B) and raw x 1 a LOWER bound (request = 64 B)
    for name in ("gather_minibatch_hot_record", "pack_hot_record", "gather_minibatch_hot_leaves", "gather_minibatch_all_leaves", "gather_minibatch_all_plain", "pack_rows",
                 "gather_minibatch_hot_plain", "loss_std_vector_1m", "loss_std_matrix_1m"):
        entry = out["cases"].get(name)
        if entry and entry["fetch_raw"] is not None and entry["write_raw"] is not None:
            lo, hi = entry["fetch_raw"] + entry["write_raw"], 2 * entry["fetch_raw"] + entry["write_raw"]
            entry["hbm_traffic_bytes_bracket"] = [lo, hi]
            entry["hbm_traffic_bytes"] = hi
            entry["traffic_over_algorithmic_bracket"] = [round(lo / entry["algorithmic_bytes"], 3), round(hi / entry["algorithmic_bytes"], 3)]
            out[name] = entry
    sq_path = directory / "SQ_counters.csv"
    if sq_path.exists():  # round 3: per-launch SQ counters of the same cases (one row per counter per dispatch)
        out["sq"] = sq_counters(sq_path, cases)
    out["ppo_loss_hip_sha256_16"] = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "ppo_loss.hip").read_bytes()).hexdigest()[:16]
    out["buffer_hip_sha256_16"] = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "buffer.hip").read_bytes()).hexdigest()[:16]
    try:
        out["commit"] = (os.environ.get("CUSRL_COMMIT")  # the GPU box gets a snapshot without .git: the caller passes the hash
                         or subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
                         or "worktree")
    except OSError:
        out["commit"] = "worktree"
    Path(out_path).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
