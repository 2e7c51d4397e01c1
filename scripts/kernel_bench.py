#!/usr/bin/env python3
"""Stand-alone timing of every HIP kernel of the hot path: back-to-back launches between one HIP-event pair
(so the stream is never host-starved), at the BASELINE config-2 size and at a roofline-scale size.

    python scripts/kernel_bench.py [--envs 4096 1048576] [--json out.json]

Prints achieved GB/s = algorithmic bytes / average launch time, and the fraction of the 8 TB/s peak.
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402

PEAK = 8000.0
DEV = "cuda:0"


USE_GRAPH = True


def timeit(fn, iters):
    """Average device time per call.  With USE_GRAPH the calls are replayed from a captured hipGraph so the Python /
    ctypes launch overhead (10-40 us per call, larger than the kernels at config-2 size) is not what is timed."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not USE_GRAPH:
        start.record()
        for _ in range(iters):
            fn()
        end.record()
        torch.cuda.synchronize()
        return start.elapsed_time(end) * 1e3 / iters  # us
    per_graph = 10
    stream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(per_graph):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    replays = max(iters // per_graph, 1)
    start.record()
    for _ in range(replays):
        graph.replay()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / (replays * per_graph)


class _Rows(dict):
    """name -> (us, bytes); measurements whose name does not match ``only`` are skipped without launching."""

    def __init__(self, only, iters):
        super().__init__()
        self.only, self.iters = only, iters

    def measure(self, name, fn, nbytes):
        only = (self.only,) if isinstance(self.only, str) else self.only
        if only is None or any(tag in name for tag in only):
            self[name] = (timeit(fn, self.iters), nbytes)


def bench_size(N, T=24, obs=48, act=12, mbs=4, only=None, iters=None):
    S = T * N
    B = S // mbs
    iters = iters or (200 if N <= 16384 else 10)
    f = lambda *shape: torch.randn(*shape, device=DEV)  # noqa: E731
    flag = lambda p: torch.rand(T, N, 1, device=DEV) < p  # noqa: E731
    rows = _Rows(only, iters)

    # ---- push: the 11 leaves of the ppo transition
    step = {"observation": f(N, obs), "mean": f(N, act), "std": f(N, act), "action": f(N, act), "logp": f(N, 1),
            "value": f(N, 1), "next_observation": f(N, obs), "reward": f(N, 1),
            "terminated": torch.rand(N, 1, device=DEV) < 0.01, "truncated": torch.rand(N, 1, device=DEV) < 0.01,
            "done": torch.rand(N, 1, device=DEV) < 0.02}
    storage = {k: torch.zeros((T,) + v.shape, dtype=v.dtype, device=DEV) for k, v in step.items()}
    pairs = [(step[k], storage[k]) for k in step]
    push_bytes = sum(2 * v.numel() * v.element_size() for v in step.values())
    cursor = [0]

    def push():
        ops.buffer_push(pairs, cursor[0], N)
        cursor[0] = (cursor[0] + 1) % T

    rows.measure("push (1 step, 11 leaves)", push, push_bytes)

    # ---- pre_update kernels
    reward, value, nv = f(T, N, 1), f(T, N, 1), f(T, N, 1)
    done, term, trunc = flag(0.015), flag(0.01), flag(0.005)
    adv, ret, out = torch.empty_like(reward), torch.empty_like(reward), torch.empty_like(reward)
    last = f(N, 1)
    rows.measure("next_value", lambda: ops.next_value(value, term, trunc, last, 0.0, False, out), S * 10)
    rows.measure("gae + return + stats", lambda: ops.gae(reward, value, nv, done, 0.99, 0.95, None, adv, ret), S * 21)
    rows.measure("gae two lambdas", lambda: ops.gae(reward, value, nv, done, 0.99, 0.95, 0.98, adv, ret), S * 21)
    mean, var = torch.zeros(1, device=DEV), torch.ones(1, device=DEV)
    rows.measure("normalize", lambda: ops.normalize_(adv, mean, var), S * 8)
    rows.measure("col_stats", lambda: ops.col_stats(adv), S * 4)
    rows.measure("compact_flags (recount)", lambda: ops.compact_flags(trunc), S * 2)

    # ---- gather: every leaf of the buffer at update time (555 B / slot)
    leaves = [f(T, N, obs), f(T, N, act), f(T, N, act), f(T, N, act), f(T, N, 1), f(T, N, 1), f(T, N, obs), f(T, N, 1),
              term, trunc, done, f(T, N, 1), f(T, N, 1), f(T, N, 1)]
    perm = torch.randperm(S, device=DEV)
    idx = perm[:B]
    row = sum(x[0, 0].numel() * x.element_size() for x in leaves)
    rows.measure(f"gather all leaves (B={B})", lambda: ops.gather_rows(leaves, idx, T, N), B * (2 * row + 8))
    small = [leaves[0], leaves[3], leaves[4], leaves[5], leaves[11], leaves[12]]
    row_small = sum(x[0, 0].numel() * x.element_size() for x in small)
    rows.measure("gather ppo-minimal leaves", lambda: ops.gather_rows(small, idx, T, N), B * (2 * row_small + 8))
    # what a captured ppo minibatch step gathers now: observation, action (plain) + old log-prob, value, advantage,
    # return, done read through the 32-byte record of the nine narrow leaves (LazyBatch + RecordPack)
    names = ["logp", "value", "reward", "terminated", "truncated", "done", "next_value", "advantage", "return"]
    narrow = dict(zip(names, [leaves[4], leaves[5], leaves[7], term, trunc, done, leaves[11], leaves[12], leaves[13]]))
    pack = ops.RecordPack(narrow)
    rows.measure("pack narrow leaves (once per update)", pack.build, S * (pack.used_bytes + pack.record_bytes))
    hot = ["logp", "value", "advantage", "return", "done"]
    hot_bytes = B * (2 * (4 * obs + 4 * act + 17) + 8)
    rows.measure(f"gather hot leaves via record (B={B})",
                 lambda: ops.gather_rows_packed([leaves[0], leaves[3]], pack, hot, idx, T, N), hot_bytes)
    hot_pack = ops.RecordPack({"observation": leaves[0], "action": leaves[3], "logp": leaves[4], "advantage": leaves[12],
                               "return": leaves[13], "done": done})
    rows.measure("pack hot record (once per update)", hot_pack.build, S * (hot_pack.used_bytes + hot_pack.record_bytes))
    # round 3: push writes the wide leaves (observation, action) through into the record, the per-update pack only moves
    # the narrow ones (13 of the 253 bytes per slot)
    rows.measure("pack the narrow leaves of the hot record (once per update, wide leaves pushed through)",
                 lambda: hot_pack.build(["logp", "advantage", "return", "done"]), S * 2 * 13)
    through_keys = ["observation", "mean", "std", "action", "logp", "value", "next_observation", "reward", "terminated", "truncated", "done"]
    through_pack = ops.RecordPack({"observation": storage["observation"], "action": storage["action"], "logp": storage["logp"],
                                   "done": storage["done"]})
    offsets = through_pack.through_offsets(through_keys)
    table = ops.make_push_table([(storage[k], step[k].shape) for k in through_keys])
    for i, k in enumerate(through_keys):
        table[i].src = step[k].data_ptr()

    def push_through():
        ops.push_table(table, len(through_keys), cursor[0], N, (through_pack.record, through_pack.record_bytes, offsets))
        cursor[0] = (cursor[0] + 1) % T

    rows.measure("push with write-through (1 step, 11 leaves + observation / action into the record)", push_through,
                 push_bytes + N * 4 * (obs + act))
    rows.measure(f"gather hot leaves from the 256 B hot record (B={B})",
                 lambda: ops.gather_rows_packed([], hot_pack, list(hot_pack.leaves), idx, T, N), B * (2 * hot_pack.used_bytes + 8))
    rows.measure(f"gather all leaves via record (B={B})",
                 lambda: ops.gather_rows_packed([leaves[0], leaves[1], leaves[2], leaves[3], leaves[6]], pack, names, idx, T, N),
                 B * (2 * row + 8))

    # ---- fused loss
    a = dict(advantage=f(B, 1), old_logp=f(B, 1) - 12, action=f(B, act), mean=f(B, act), std=torch.rand(B, act, device=DEV) + 0.5,
             ret=f(B, 1), curr_value=f(B, 1), old_value=f(B, 1))
    kw = dict(clip=0.2, value_clip=None, w_sur=1.0, w_val=0.5, w_ent=0.01)
    loss_bytes = B * (8 + 3 * 4 * act + 8 + 2 * 4 * act + 4 + 16)
    rows.measure(f"ppo loss fwd+bwd (B={B})", lambda: ops.ppo_loss_fwd_bwd(*a.values(), **kw), loss_bytes)
    if ops.ppo_loss_accepts_std_vector(act):  # state-independent std handed over as its [A] vector (the ppo preset's case)
        v = dict(a, std=torch.rand(act, device=DEV) + 0.5)
        rows.measure(f"ppo loss fwd+bwd, std vector (B={B})", lambda: ops.ppo_loss_fwd_bwd(*v.values(), **kw), loss_bytes - B * 8 * act)

        # round 6: what a captured step with the critic on its own stream launches instead — the surrogate + entropy terms
        # without the value term (actor's stream) and the value term alone (critic's stream)
        policy_only = dict(v, ret=None, curr_value=None, old_value=None)
        rows.measure(f"ppo loss fwd+bwd, std vector, no value term (B={B})", lambda: ops.ppo_loss_fwd_bwd(*policy_only.values(), **kw),
                     loss_bytes - B * 8 * act - B * 12)
    rows.measure(f"value term fwd+bwd (B={B})",
                 lambda: ops.value_loss_fwd_bwd(a["ret"], a["curr_value"], None, value_clip=None, w_val=0.5), B * 12)

    # ---- MLP backward epilogues and the optimizer-side kernels of one minibatch step
    g256, y256 = f(B, 256), torch.relu(f(B, 256))
    rows.measure(f"relu bwd + bias grad [B,256] (B={B})", lambda: ops.relu_backward_bias(g256, y256), B * 256 * 12)
    # round 6: the bottom layer's whole backward (mask + bias gradient + weight gradient, nothing written back per row)
    x_obs = f(B, obs)
    if ops.input_layer_supported(g256, y256, x_obs, torch.empty(256, obs, device=DEV)):
        rows.measure(f"input layer bwd [B,{obs}]x[B,256] (B={B})", lambda: ops.input_layer_backward(g256, y256, x_obs),
                     B * 4 * (2 * 256 + obs))
    gm, gv, h2 = f(B, act), f(B, 1), f(B, 128)
    wm, wv = f(act, 128), f(1, 128)
    rows.measure(f"narrow head bwd 128->{act} (B={B})", lambda: ops.narrow_linear_backward(gm, h2, wm), B * 4 * (act + 256))
    rows.measure(f"narrow head bwd 128->1 (B={B})", lambda: ops.narrow_linear_backward(gv, h2, wv), B * 4 * (1 + 256))
    flat = f(92569)
    rows.measure("clip_grad_norm (92569 floats)", lambda: ops.clip_grad_norm_(flat, 1.0), flat.numel() * 12)

    # ---- recurrent backbone (config 4): one GRU time step's gate pass over N sequences of 256 hidden units
    Hs = 256
    gi, gh, bh = f(N, 3 * Hs), f(N, 3 * Hs), f(3 * Hs)
    hstate, hout, hprev, dh, dout = f(N, Hs), f(N, Hs), f(N, Hs), f(N, Hs), f(N, Hs)
    rows.measure(f"gru gates fwd [B,256] (B={N})", lambda: ops.gru_gates_forward(gi, gh, bh, hstate, hout, None, 0), N * Hs * 36)
    rows.measure(f"gru gates bwd [B,256] (B={N})", lambda: ops.gru_gates_backward(gi, gh, bh, hprev, dout, dh, None, 0), N * Hs * 64)
    logits3, race3 = f(N, 3), torch.rand(N, 3, device=DEV) + 0.1
    rows.measure(f"categorical sample + logp [B,3] (B={N})", lambda: ops.categorical_sample_logp(logits3, race3), N * (3 * 12 + 4))

    # ---- reference points: a plain device copy of the same bytes (what the memory system gives a streaming kernel)
    big = torch.empty(max(S * 21 // 8, 1024), dtype=torch.float32, device=DEV)
    dst = torch.empty_like(big)
    rows.measure("torch copy_ (same bytes as gae)", lambda: dst.copy_(big), big.numel() * 8)
    return rows


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, nargs="+", default=[4096, 1048576])
    parser.add_argument("--json", type=str, default=None)
    parser.add_argument("--only", type=str, nargs="+", default=None, help="substring filter(s) on row names")
    parser.add_argument("--iters", type=int, default=None)
    parser.add_argument("--mbs", type=int, default=4, help="minibatches per epoch (1 = gather the whole buffer)")
    parser.add_argument("--eager", action="store_true", help="time eager launches (includes host launch overhead)")
    args = parser.parse_args()
    global USE_GRAPH
    USE_GRAPH = not args.eager
    report = {}
    for N in args.envs:
        rows = bench_size(N, mbs=args.mbs, only=None if args.only is None else tuple(args.only), iters=args.iters)
        print(f"\n== N = {N} envs, T = 24 ==")
        print(f"{'kernel':42s} {'us/launch':>10s} {'MB':>9s} {'GB/s':>9s} {'% of 8TB/s':>10s}")
        report[str(N)] = {}
        for name, (us, nbytes) in rows.items():
            gbs = nbytes / us / 1e3
            print(f"{name:42s} {us:10.2f} {nbytes / 1e6:9.2f} {gbs:9.1f} {100 * gbs / PEAK:9.1f}%")
            report[str(N)][name] = {"us": round(us, 3), "bytes": int(nbytes), "GBps": round(gbs, 1), "frac": round(gbs / PEAK, 4)}
    if args.json:
        Path(args.json).write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
