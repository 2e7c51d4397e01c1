#!/usr/bin/env python3
"""``python scripts/pmc_r06_summarize.py <dir> <out.json>``: the PMC passes of scripts/pmc_r06_cases.py per case — read counter
x its calibrated streaming factor + write counter = memory-side traffic per launch, next to the launch's algorithmic bytes;
SQ shares where the pass exists.  Counters come in KB (rocprofv3 derived metrics)."""
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def rows_of(path, kernel):
    with open(path) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda row: int(row["Dispatch_Id"]))
    return [row for row in rows if kernel in row["Kernel_Name"]]


def main(directory, out_path):
    directory = Path(directory)
    cases = json.loads((directory / "cases.json").read_text())
    out = {"unit": "bytes per launch", "cases": {}}

    def average(counter_file, case):
        rows = rows_of(directory / counter_file, case["kernel"])
        if "grid_threads" in case:
            rows = [row for row in rows if int(row["Grid_Size"]) == case["grid_threads"]]
        values = [float(row["Counter_Value"]) * 1024 for row in rows]
        return (sum(values) / len(values) if values else None), len(values)

    stream_fetch, _ = average("FETCH_SIZE_counters.csv", cases["stream_16B"])
    factor = (1 << 30) / stream_fetch if stream_fetch else 2.0
    out["calibration"] = {"stream_16B_read_factor": factor}
    for name, case in cases.items():
        fetch, n_fetch = average("FETCH_SIZE_counters.csv", case)
        write, n_write = average("WRITE_SIZE_counters.csv", case)
        entry = {"algorithmic_bytes": case["algorithmic_bytes"], "fetch_raw": fetch, "write_raw": write, "rows_matched": [n_fetch, n_write]}
        if fetch is not None and write is not None:
            entry["hbm_traffic_bytes"] = fetch * factor + write
            entry["traffic_over_algorithmic"] = round(entry["hbm_traffic_bytes"] / case["algorithmic_bytes"], 3)
        if "partial_bytes" in case:
            entry["partial_row_bytes_written_and_read_back"] = case["partial_bytes"]
        out["cases"][name] = entry
    sq = directory / "SQ_counters.csv"
    if sq.exists():
        out["sq"] = {}
        for name, case in cases.items():
            if "grid_threads" not in case:
                continue
            per_dispatch = {}
            for row in rows_of(sq, case["kernel"]):
                if int(row["Grid_Size"]) == case["grid_threads"]:
                    per_dispatch.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
            if per_dispatch:
                keys = next(iter(per_dispatch.values())).keys()
                mean = {key: sum(d.get(key, 0.0) for d in per_dispatch.values()) / len(per_dispatch) for key in keys}
                wave = mean.get("SQ_WAVE_CYCLES") or 0.0
                if wave:
                    mean["valu_issue_share_of_wave_cycles"] = round(mean.get("SQ_ACTIVE_INST_VALU", 0.0) / wave, 4)
                    mean["waiting_share_of_wave_cycles"] = round(mean.get("SQ_WAIT_ANY", 0.0) / wave, 4)
                out["sq"][name] = mean
    out["input_layer_hip_sha256_16"] = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "input_layer.hip").read_bytes()).hexdigest()[:16]
    Path(out_path).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
