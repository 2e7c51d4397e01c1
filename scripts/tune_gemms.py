#!/usr/bin/env python3
"""Pick the fastest rocBLAS / hipBLASLt kernel for every GEMM shape of the `ppo` preset's minibatch step and rollout
(PyTorch TunableOp) and write the selection to cusrl_amd/tuned_gemms_gfx950.csv.  The MLP GEMMs stay library GEMMs
(BASELINE.json north_star); this only replaces the libraries' default heuristic choice by a measured one.

    python scripts/tune_gemms.py [--envs 4096 8192] [--max-ms 10]
    python scripts/tune_gemms.py --envs --configs config5 config4 config1 --out gpurun_out/tuned_new.csv
    python scripts/merge_tuned_gemms.py gpurun_out/tuned_new.csv      # add the new shapes to the shipped selection

Runs the preset eagerly (tuning cannot happen inside hipGraph capture; the shapes are the same) for two iterations per
size.  `import cusrl_amd` loads the file when it matches the installed ROCm / hipBLASLt (TunableOp validates that)."""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, nargs="*", default=[4096])
    parser.add_argument("--configs", nargs="*", default=[], help="BASELINE configs of scripts/run_config.py to tune as well "
                        "(their per-GPU workloads: RND / AMP networks, the recurrent cores, the discrete toy)")
    parser.add_argument("--max-ms", type=int, default=30)
    parser.add_argument("--max-iterations", type=int, default=100)
    parser.add_argument("--out", type=str, default=str(ROOT / "cusrl_amd" / "tuned_gemms_gfx950.csv"))
    args = parser.parse_args()
    import os

    os.environ["CUSRL_TUNED_GEMMS"] = "0"  # start from the libraries' defaults, not from an earlier selection
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(args.max_ms)
    tunable.set_max_tuning_iterations(args.max_iterations)
    tunable.set_filename(args.out)
    import cusrl_amd as cusrl  # noqa: E402

    cusrl.config.set_device("cuda:0")
    for envs in args.envs:
        cusrl.set_global_seed(42)
        env = cusrl.testing.SyntheticEnvironment(envs, 48, 12, device="cuda:0")
        factory = cusrl.preset.PpoAgentFactory(optimizer_kwargs={"capturable": True, "fused": True})
        trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
        observation, state, _ = env.reset()
        t0 = time.perf_counter()
        for _ in range(2):
            observation, state = trainer._rollout_and_update(observation, state)
        torch.cuda.synchronize()
        print(f"{envs} envs: tuned in {time.perf_counter() - t0:.1f} s, {len(tunable.get_results())} entries", flush=True)
    sys.path.insert(0, str(ROOT / "scripts"))
    import run_config  # noqa: E402

    for name in args.configs:
        cusrl.set_global_seed(42)
        env, factory = run_config.build(name, None, False)
        trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
        observation, state, _ = env.reset()
        t0 = time.perf_counter()
        for _ in range(2):
            observation, state = trainer._rollout_and_update(observation, state)
        torch.cuda.synchronize()
        print(f"{name}: tuned in {time.perf_counter() - t0:.1f} s, {len(tunable.get_results())} entries", flush=True)
    print("validators:", tunable.get_validators())
    print("results go to", args.out, "at interpreter exit (TunableOp writes its file then)")


if __name__ == "__main__":
    main()
