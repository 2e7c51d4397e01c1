#!/usr/bin/env python3
"""Average every counter of scripts/gpu_pmc_r04.sh's passes over the launches of each case of scripts/pmc_r04_cases.py.

    python scripts/pmc_r04_summarize.py <dir with cases.json and *_counters.csv> <out.json>

Launches are matched by kernel name + ordinal in dispatch order.  Derived figures (all per launch):
  fetch_bytes      FETCH_SIZE (KB) x 1024 x 2 — the gfx950 correction for 16 B/lane streaming reads (MI355X_MICROARCH.md
                   §HBM; calibrated in profiles/r02/pmc_summary.json: factor 2.000)
  write_bytes      WRITE_SIZE (KB) x 1024
  traffic_ratio    (fetch_bytes + write_bytes) / algorithmic bytes
  l2_hit_rate      TCC_HIT / (TCC_HIT + TCC_MISS)
  wait_share       SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt), issue_share = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  read_latency     TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles per L1->L2 read request), same for writes
"""
import csv
import json
import sys
from pathlib import Path


def main(directory, out_path):
    directory = Path(directory)
    cases = json.loads((directory / "cases.json").read_text())
    kernels = sorted({case["kernel"] for case in cases.values()})
    summary = {name: dict(case, counters={}) for name, case in cases.items()}
    for path in sorted(directory.glob("*_counters.csv")):
        with open(path) as fh:
            rows = list(csv.DictReader(fh))
        dispatches: dict[int, tuple[str, dict]] = {}
        for row in rows:
            kernel = next((k for k in kernels if f"cusrl::{k}" in row["Kernel_Name"]), None)
            if kernel is None:
                continue
            entry = dispatches.setdefault(int(row["Dispatch_Id"]), (kernel, {}))
            entry[1][row["Counter_Name"]] = entry[1].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        per_kernel: dict[str, list[dict]] = {}
        for dispatch in sorted(dispatches):
            kernel, counters = dispatches[dispatch]
            per_kernel.setdefault(kernel, []).append(counters)
        for name, case in cases.items():
            mine = per_kernel.get(case["kernel"], [])[case["first_ordinal"]: case["first_ordinal"] + case["launches"]]
            if not mine:
                continue
            for key in mine[0]:
                summary[name]["counters"][key] = sum(c.get(key, 0.0) for c in mine) / len(mine)
    for name, case in summary.items():
        c, d = case["counters"], {}
        if "FETCH_SIZE" in c:
            d["fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in c:
            d["write_bytes"] = c["WRITE_SIZE"] * 1024
        if "fetch_bytes" in d and "write_bytes" in d:
            d["traffic_ratio"] = round((d["fetch_bytes"] + d["write_bytes"]) / case["algorithmic_bytes"], 3)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if c.get("SQ_WAVE_CYCLES"):
            d["wait_share"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            d["issue_share"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            d["issue_stall_share"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_ACTIVE_INST_VALU") is not None:
            d["valu_share"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3)
            d["lds_share"] = round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            d["vmem_share"] = round(c.get("SQ_ACTIVE_INST_VMEM", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            d["lds_issue_stall_share"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c.get("TCP_TCC_READ_REQ_sum"):
            d["read_latency_cycles"] = round(c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / c["TCP_TCC_READ_REQ_sum"], 1)
        if c.get("TCP_TCC_WRITE_REQ_sum"):
            d["write_latency_cycles"] = round(c.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0.0) / c["TCP_TCC_WRITE_REQ_sum"], 1)
        if c.get("TCC_EA0_RDREQ_sum"):
            d["ea_read_dram_share"] = round(c.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / c["TCC_EA0_RDREQ_sum"], 3)
        if c.get("TCC_EA0_WRREQ_sum"):
            d["ea_write_dram_share"] = round(c.get("TCC_EA0_WRREQ_DRAM_sum", 0.0) / c["TCC_EA0_WRREQ_sum"], 3)
        if c.get("TCC_BUSY_sum"):
            d["ea_wrreq_stall_per_busy"] = round(c.get("TCC_EA0_WRREQ_STALL_sum", 0.0) / c["TCC_BUSY_sum"], 4)
            d["too_many_wrreqs_stall_per_busy"] = round(c.get("TCC_TOO_MANY_EA_WRREQS_STALL_sum", 0.0) / c["TCC_BUSY_sum"], 4)
            d["tag_stall_per_busy"] = round(c.get("TCC_TAG_STALL_sum", 0.0) / c["TCC_BUSY_sum"], 4)
        case["derived"] = d
    Path(out_path).write_text(json.dumps(summary, indent=1))
    for name, case in summary.items():
        print(f"{name:34s} {json.dumps(case['derived'])}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
