#!/usr/bin/env python3
"""Add the entries of a fresh TunableOp result file to cusrl_amd/tuned_gemms_gfx950.csv.

    python scripts/merge_tuned_gemms.py gpurun_out/tuned_new.csv [--replace]

Entries already shipped win unless --replace is given (the headline's selection stays the one its numbers were measured
with); the validator lines must agree (same PyTorch / ROCm / hipBLASLt / rocBLAS), and shapes with a data-dependent row count — the eager
truncated-state bootstrap, the packed / per-time-step row counts of the recurrent update — are dropped: kept are row
counts that are a multiple of 256 (envs, envs x steps, minibatches, the bucketed recurrent steps) or a multiple of 8 up to 512 (the toy config, the
AMP discriminator batch, the weight-gradient slabs)."""
import argparse
import re
import sys
from pathlib import Path

SHIPPED = Path(__file__).resolve().parent.parent / "cusrl_amd" / "tuned_gemms_gfx950.csv"


def parse(path):
    validators, entries = [], {}
    for line in Path(path).read_text().splitlines():
        if not line.strip():
            continue
        if line.startswith("Validator,"):
            validators.append(line)
        else:
            op, shape, rest = line.split(",", 2)
            entries[(op, shape)] = rest
    return validators, entries


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("new")
    parser.add_argument("--replace", action="store_true")
    args = parser.parse_args()
    validators, shipped = parse(SHIPPED)
    new_validators, fresh = parse(args.new)
    if sorted(validators) != sorted(new_validators):
        sys.exit(f"validator lines differ:\n{validators}\n{new_validators}")
    added = replaced = dropped = 0
    for key, rest in fresh.items():
        rows = int(re.match(r"[a-z]+_(\d+)_(\d+)_(\d+)", key[1]).group(2))
        if rows % 256 and (rows % 8 or rows > 512):
            dropped += 1
            continue
        if key in shipped and not args.replace:
            continue
        replaced += key in shipped
        added += key not in shipped
        shipped[key] = rest
    SHIPPED.write_text("\n".join(validators + [",".join((*key, rest)) for key, rest in shipped.items()]) + "\n")
    print(f"{added} added, {replaced} replaced, {dropped} dropped (data-dependent row counts); {len(shipped)} entries shipped")


if __name__ == "__main__":
    main()
