#!/usr/bin/env python3
"""Round-5 PMC cases: the fused PPO objective at roofline scale (1 048 576 envs x 24 steps / 4 minibatches = 6 291 456 rows,
A = 12), each launched a few times under ``rocprofv3 --pmc <one counter set> --kernel-trace`` (scripts/gpu_pmc_r05.sh runs one
pass per set; scripts/pmc_r04_summarize.py averages the launches of each case):

  *_per_round   round 4's kernel: the per-row scalar streams moved once per ROUND (CUSRL_LOSS_WAVE_ROWS=0), default cache policy
  *_per_wave    round 5's scalar-stream layout (once per WAVE), default cache policy (CUSRL_LOSS_POLICY=0)
  *_streaming   what ships at this size: the [B, A] streams non-temporal, loads and stores (CUSRL_LOSS_POLICY=1)

  loss_std_vector_1m_*   cusrl::ppo_loss_rowgroup_kernel<3, true, true, ...>  (the preset's form, 180 B / row)
  loss_std_matrix_1m_*   cusrl::ppo_loss_rowgroup_kernel<3, false, true, ...> (276 B / row)
  loss_config2_*         the in-step launch of config 2 (24 576 rows: latency-bound, cache-resident; default policy by footprint)
"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402

DEV = "cuda:0"
REPEAT = 4


def main(out_dir):
    cases, issued = {}, {}
    f = lambda *shape: torch.randn(*shape, device=DEV)  # noqa: E731

    def run(name, kernel, fn, algorithmic_bytes, env):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            fn()
            torch.cuda.synchronize()
            first = issued.get(kernel, 0) + 1
            for _ in range(REPEAT):
                fn()
                torch.cuda.synchronize()
            issued[kernel] = first + REPEAT
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        cases[name] = {"kernel": kernel, "first_ordinal": first, "launches": REPEAT, "algorithmic_bytes": int(algorithmic_bytes),
                       "cold": False}

    act = 12
    kw = dict(clip=0.2, value_clip=None, w_sur=1.0, w_val=0.5, w_ent=0.01)
    for tag, B in (("1m", (1 << 20) * 24 // 4), ("config2", 24576)):
        a = dict(advantage=f(B, 1), old_logp=f(B, 1) - 12, action=f(B, act), mean=f(B, act), std=torch.rand(B, act, device=DEV) + 0.5,
                 ret=f(B, 1), curr_value=f(B, 1), old_value=f(B, 1))
        v = dict(a, std=torch.rand(act, device=DEV) + 0.5)
        matrix_bytes = B * (8 + 3 * 4 * act + 8 + 2 * 4 * act + 4 + 16)
        variants = (("per_round", "0", "0"), ("per_wave", "1", "0")) + ((("streaming", "1", "1"),) if tag == "1m" else ())
        for layout, flag, policy in variants:
            env = {"CUSRL_LOSS_WAVE_ROWS": flag, "CUSRL_LOSS_POLICY": policy}
            if tag == "1m":
                run(f"loss_std_matrix_{tag}_{layout}", "ppo_loss_rowgroup_kernel", lambda: ops.ppo_loss_fwd_bwd(*a.values(), **kw), matrix_bytes, env)
            run(f"loss_std_vector_{tag}_{layout}" if tag == "1m" else f"loss_{tag}_{layout}", "ppo_loss_rowgroup_kernel",
                lambda: ops.ppo_loss_fwd_bwd(*v.values(), **kw), matrix_bytes - B * 8 * act, env)
        del a, v
        torch.cuda.empty_cache()
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    (Path(out_dir) / "cases.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
