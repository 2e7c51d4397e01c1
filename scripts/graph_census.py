#!/usr/bin/env python3
"""Node census of the hipGraphs a run captured (``cusrl_graph_census`` through ``template/graphs.py::_Capture.censuses``):
per captured region the number of kernel / memcpy / memset nodes and the kernel families (cusrl / GEMM / ATen reduce /
other ATen).  A captured minibatch step must contain NO memset node and NO ATen ``reduce_kernel``.

    python scripts/graph_census.py config2 [--envs N] [--iterations K]     # runs scripts/run_config.py's workload, compiled
"""
import re
import sys
from collections import Counter
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def family(name: str) -> str:
    if "reduce_kernel" in name:
        return "aten_reduce"
    if "cusrl" in name:
        return "cusrl"
    if name.startswith("Cijk_") or "Cijk" in name or "rocblas" in name.lower() or "hipblaslt" in name.lower():
        return "gemm"
    if "2at6native" in name or "at::native" in name:
        return "aten_other"
    return "other"


def summarize(censuses, out=sys.stdout) -> dict:
    """Prints one line per distinct (region, node signature); returns the totals."""
    seen = Counter()
    lines = {}
    totals = Counter()
    for c in censuses:
        fam = Counter(family(n) for n in c["names"])
        key = (c.get("region", "?"), c["kernel"], c["memcpy"], c["memset"], tuple(sorted(fam.items())))
        seen[key] += 1
        lines[key] = fam
        totals["memset"] += c["memset"]
        totals["aten_reduce"] += fam.get("aten_reduce", 0)
        totals["graphs"] += 1
    for key, count in seen.items():
        region, kernel, memcpy, memset, _ = key
        print(f"{count:4d} x {region}: kernel {kernel} memcpy {memcpy} memset {memset} families {dict(lines[key])}", file=out)
    print(f"total: {totals['graphs']} captured graphs, {totals['memset']} memset nodes, {totals['aten_reduce']} ATen reduce_kernel nodes", file=out)
    return dict(totals)


def aten_names(censuses) -> Counter:
    names = Counter()
    for c in censuses:
        for n in c["names"]:
            if family(n) in ("aten_reduce", "aten_other"):
                names[re.sub(r"\s+", " ", n)[:140]] += 1
    return names


def main():
    import argparse

    import torch

    import cusrl_amd as cusrl
    from cusrl_amd.template.graphs import _Capture
    from run_config import build

    parser = argparse.ArgumentParser()
    parser.add_argument("config")
    parser.add_argument("--envs", type=int, default=None)
    parser.add_argument("--iterations", type=int, default=4)
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env, factory = build(args.config, args.envs, True)
    trainer = cusrl.Trainer(env, factory, num_iterations=args.iterations, verbose=False)
    trainer.run_training_loop()
    torch.cuda.synchronize()
    summarize(_Capture.censuses)
    for name, count in aten_names(_Capture.censuses).most_common(30):
        print(f"     {count:5d} x {name}")


if __name__ == "__main__":
    main()
