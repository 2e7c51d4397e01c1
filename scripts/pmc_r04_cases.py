#!/usr/bin/env python3
"""Round-4 PMC cases: the pre_update kernels and the push at roofline scale (1 048 576 envs x 24 steps), each launched a
few times under ``rocprofv3 --pmc <one counter set> --kernel-trace`` (scripts/gpu_pmc_r04.sh runs one pass per set).

A case owns ``REPEAT`` consecutive launches of one kernel; the summariser (scripts/pmc_r04_summarize.py) finds them by kernel
name + ordinal in dispatch order, which this script records in ``<out>/cases.json`` with the algorithmic bytes per launch.

  gae_1m_default_policy_{hot,cold}   cusrl::gae_kernel, every stream with the default cache policy (round 3's launch)
  gae_1m_streaming_{hot,cold}        what ships: non-temporal loads, `return` stored non-temporally, advantage kept cached
  next_value_1m, normalize_1m        the launches in front of and behind the scan
  push_1m                            one step of all 11 leaves of the ppo transition (1.14 GB)
  gae_config2                        the config-2 launch (4096 envs: latency-bound, cache-resident)
  narrow_head_bwd_12_config2         cusrl::narrow_linear_bwd_kernel<12> at B = 24 576 (0.20 of the roofline: counters first)
  relu_bwd_colsum_256_config2        cusrl::colsum_chunked_kernel<true> at [24 576, 256] (the largest cusrl:: kernel of config 2)

hot  = launches back to back.  cold = 1 GiB of fresh writes in front of every launch (the cache state a kernel meets in
an update: full of somebody else's dirty lines).
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
    scratch = torch.empty(1 << 28, device=DEV)

    def run(name, kernel, fn, algorithmic_bytes, cold=False, env=None):
        saved = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            fn()  # warm (not counted as part of the case)
            torch.cuda.synchronize()
            first = issued.get(kernel, 0) + 1
            for _ in range(REPEAT):
                if cold:
                    scratch.fill_(1.0)
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
                       "cold": cold}

    T, N = 24, 1 << 20
    S = T * N
    reward, value, nv, last = f(T, N, 1), f(T, N, 1), f(T, N, 1), f(N, 1)
    term, trunc = torch.rand(T, N, 1, device=DEV) < 0.01, torch.rand(T, N, 1, device=DEV) < 0.005
    done = term | trunc
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    gae = lambda: ops.gae(reward, value, nv, done, 0.99, 0.95, None, adv, ret)  # noqa: E731
    default = {"CUSRL_GAE_POLICY": "0", "CUSRL_GAE_BLOCK": "256"}
    for cold in (False, True):
        tag = "cold" if cold else "hot"
        run(f"gae_1m_default_policy_{tag}", "gae_kernel", gae, 21 * S, cold, default)
        run(f"gae_1m_streaming_{tag}", "gae_kernel", gae, 21 * S, cold)
    run("next_value_1m_hot", "next_value_kernel", lambda: ops.next_value(value, term, trunc, last, 0.0, False, nv), 10 * S)
    run("next_value_1m_cold", "next_value_kernel", lambda: ops.next_value(value, term, trunc, last, 0.0, False, nv), 10 * S, True)
    partials = ops.gae(reward, value, nv, done, 0.99, 0.95, None, adv, ret)[2]
    issued["gae_kernel"] = issued.get("gae_kernel", 0) + 1
    run("normalize_1m_hot", "normalize_from_partials_kernel", lambda: ops.normalize_from_partials_(adv, partials, S), 8 * S)
    run("normalize_1m_cold", "normalize_from_partials_kernel", lambda: ops.normalize_from_partials_(adv, partials, S), 8 * S, True)
    del reward, value, nv, adv, ret, term, trunc, done
    torch.cuda.empty_cache()

    obs, act = 48, 12
    step = {"observation": f(N, obs), "mean": f(N, act), "std": f(N, act), "action": f(N, act), "logp": f(N, 1), "value": f(N, 1),
            "next_observation": f(N, obs), "reward": f(N, 1), "terminated": torch.rand(N, 1, device=DEV) < 0.01,
            "truncated": torch.rand(N, 1, device=DEV) < 0.01, "done": torch.rand(N, 1, device=DEV) < 0.02}
    storage = {k: torch.zeros((2,) + v.shape, dtype=v.dtype, device=DEV) for k, v in step.items()}
    pairs = [(step[k], storage[k]) for k in step]
    push_bytes = sum(2 * v.numel() * v.element_size() for v in step.values())
    cursor = [0]

    def push():
        ops.buffer_push(pairs, cursor[0], N)
        cursor[0] ^= 1

    run("push_1m_hot", "push_kernel", push, push_bytes)
    run("push_1m_cold", "push_kernel", push, push_bytes, True)
    del step, storage, pairs
    torch.cuda.empty_cache()

    T, N = 24, 4096
    S = T * N
    reward, value, nv = f(T, N, 1), f(T, N, 1), f(T, N, 1)
    done = torch.rand(T, N, 1, device=DEV) < 0.015
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    run("gae_config2", "gae_kernel", lambda: ops.gae(reward, value, nv, done, 0.99, 0.95, None, adv, ret), 21 * S)
    # ---- the two largest non-GEMM kernels of a config-2 minibatch step (callers of the path: MLP backward epilogues)
    B = 24576
    g12, h128, w12 = f(B, 12), torch.relu(f(B, 128)), f(12, 128)
    run("narrow_head_bwd_12_config2", "narrow_linear_bwd_kernel", lambda: ops.narrow_linear_backward(g12, h128, w12), B * 4 * (12 + 256))
    g256, y256 = f(B, 256), torch.relu(f(B, 256))
    run("relu_bwd_colsum_256_config2", "colsum_chunked_kernel", lambda: ops.relu_backward_bias(g256, y256), B * 256 * 12)
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    Path(out_dir, "cases.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
