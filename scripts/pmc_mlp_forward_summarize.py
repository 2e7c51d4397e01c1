#!/usr/bin/env python3
"""``python scripts/pmc_mlp_forward_summarize.py <dir> <out.json>``: the PMC passes of scripts/pmc_mlp_forward_cases.py per case —
memory-side traffic (read counter x its calibrated streaming factor + write counter) next to the algorithmic bytes, and the SQ
pass: MFMA instructions issued against the count the launch must issue, the matrix cores' busy share of the SQ's busy cycles."""
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
    out = {"unit": "per launch", "cases": {}}

    def average(counter_file, case):
        rows = rows_of(directory / counter_file, case["kernel"])
        if "grid_threads" in case:
            rows = [row for row in rows if int(row["Grid_Size"]) == case["grid_threads"]]
        if "order" in case:  # the cases share a grid: this case's launches are the order-th run of `launches` dispatches
            rows = rows[case["order"] * case["launches"] : (case["order"] + 1) * case["launches"]]
        values = [float(row["Counter_Value"]) * 1024 for row in rows]
        return (sum(values) / len(values) if values else None), len(values)

    stream_fetch, _ = average("FETCH_SIZE_counters.csv", cases["stream_16B"])
    factor = (1 << 30) / stream_fetch if stream_fetch else 2.0
    out["calibration"] = {"stream_16B_read_factor": factor}
    for name, case in cases.items():
        fetch, n_fetch = average("FETCH_SIZE_counters.csv", case)
        write, n_write = average("WRITE_SIZE_counters.csv", case)
        entry = {key: case[key] for key in ("rows", "flops", "mfma_instructions", "algorithmic_bytes") if key in case}
        if fetch is not None and write is not None:
            entry["hbm_traffic_bytes"] = fetch * factor + write
            entry["traffic_over_algorithmic"] = round(entry["hbm_traffic_bytes"] / case["algorithmic_bytes"], 3)
        entry["rows_matched"] = [n_fetch, n_write]
        sq = directory / "SQ_counters.csv"
        if sq.exists() and "grid_threads" in case:
            per_dispatch = {}
            for row in rows_of(sq, case["kernel"]):
                if int(row["Grid_Size"]) == case["grid_threads"]:
                    per_dispatch.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
            if "order" in case:
                ids = sorted(per_dispatch)[case["order"] * case["launches"] : (case["order"] + 1) * case["launches"]]
                per_dispatch = {i: per_dispatch[i] for i in ids}
            if per_dispatch:
                keys = next(iter(per_dispatch.values())).keys()
                mean = {key: sum(d.get(key, 0.0) for d in per_dispatch.values()) / len(per_dispatch) for key in keys}
                entry["sq"] = mean
                if mean.get("SQ_BUSY_CYCLES"):
                    entry["mfma_busy_share_of_sq_busy_cycles"] = round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / mean["SQ_BUSY_CYCLES"], 4)
                if "mfma_instructions" in case and mean.get("SQ_INSTS_MFMA"):
                    entry["mfma_issued_over_required"] = round(mean["SQ_INSTS_MFMA"] / case["mfma_instructions"], 4)
                    # v_mfma_f32_16x16x4_f32 occupies its SIMD's matrix pipe for 32 cycles (2048 flop at 64 flop / clk / SIMD)
                    entry["mfma_busy_cycles_per_instruction"] = round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / mean["SQ_INSTS_MFMA"], 2)
                if mean.get("SQ_BUSY_CYCLES") and mean.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    # the busy counter sums over the 1024 SIMDs (256 CUs x 4), SQ_BUSY_CYCLES over the 32 shader engines (8 XCDs x 4):
                    # matrix-pipe busy cycles per SIMD over the launch's cycles = the share of the fp32 MFMA peak the launch ran at
                    entry["mfma_pipe_utilisation"] = round((mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (mean["SQ_BUSY_CYCLES"] / 32), 4)
                    entry["launch_cycles_from_sq_busy"] = round(mean["SQ_BUSY_CYCLES"] / 32)
        out["cases"][name] = entry
    out["mlp_forward_hip_sha256_16"] = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "mlp_forward.hip").read_bytes()).hexdigest()[:16]
    Path(out_path).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
