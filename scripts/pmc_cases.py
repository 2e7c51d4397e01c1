#!/usr/bin/env python3
"""Launches whose HBM traffic the PMC passes measure (run under ``rocprofv3 --pmc FETCH_SIZE`` and, separately,
``--pmc WRITE_SIZE`` by scripts/gpu_pmc.sh).  Every case is one C-ABI launch repeated a few times eagerly; the cases are
told apart afterwards by kernel name + grid size, which this script writes to ``<out>/cases.json`` together with the
algorithmic bytes of each launch.

Calibration cases (sizes far beyond the 256 MB Infinity Cache, distinct rows, so every request reaches HBM):
  stream_16B        push of one 1 GiB leaf: 16 B/lane streaming read — the access the guide's x2 FETCH_SIZE factor is for
  random_rows_4B    gather of 8 Mi distinct random rows of a 4-byte-row leaf  (narrow leaf at random slots)
  random_rows_32B   gather of 8 Mi distinct random rows of a 32-byte-row leaf (what the packed record turns them into)
  random_rows_192B  gather of 2 Mi distinct random rows of a 192-byte-row leaf (an observation row)
Product cases at config 2 (4096 envs x 24 steps, minibatch of 24 576 slots; working set L2 / MALL resident):
  gather_minibatch_hot_record   what the captured train step launches: 6 leaves out of the 256-byte hot record
  gather_minibatch_hot_leaves   the same leaves + value with only the narrow ones in a (32-byte) record
  gather_minibatch_all_leaves   the reference's semantics: all 14 leaves (9 through the record)
  gather_minibatch_all_plain    all 14 leaves without the record (the round-1 kernel's access pattern)
  pack_rows                     building the record (once per update)
  gather_minibatch_hot_plain    (round 3) what the captured train step launches at config 2 now: the six leaves a PPO step
                                reads, gathered plainly — the 25 MB of sampled leaves sit in L2 + Infinity Cache
Loss kernel at 1 048 576 envs (minibatch of 6 291 456 rows x 12 actions; round 3, cusrl::ppo_loss_rowgroup_kernel):
  loss_std_vector_1m            the preset's form: std handed over as its [12] vector (180 algorithmic bytes per row)
  loss_std_matrix_1m            std as a [B, 12] matrix (276 bytes per row)
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402

DEV = "cuda:0"
REPEAT = 5
KBLOCK, ITEMS = 256, 4


def blocks_plain(rows, row_bytes, unit):
    return -(-(rows * (row_bytes // unit)) // (KBLOCK * ITEMS))


def main(out_dir):
    cases = {}
    issued = {}  # launches of each of our kernels so far: a case owns the ordinals [first, first + REPEAT * per_call)
    f = lambda *shape: torch.randn(*shape, device=DEV)  # noqa: E731

    def run(name, fn, kernel, grid_blocks, algorithmic_bytes, launches_per_call=None):
        per_call = launches_per_call or {kernel: 1}
        first = issued.get(kernel, 0)
        for _ in range(REPEAT):
            fn()
        torch.cuda.synchronize()
        for k, n in per_call.items():
            issued[k] = issued.get(k, 0) + n * REPEAT
        cases[name] = {"kernel": kernel, "grid_threads": int(grid_blocks) * KBLOCK, "algorithmic_bytes": int(algorithmic_bytes),
                       "launches": REPEAT, "first_ordinal": first, "ordinals": per_call[kernel] * REPEAT}

    # ---- calibration
    big = torch.empty(1, 1 << 26, 4, device=DEV).normal_()  # [T=1, N=64 Mi, 4 floats] = 1 GiB, 16-byte rows
    step = big[0]
    storage = torch.empty_like(big)
    run("stream_16B", lambda: ops.buffer_push([(step, storage)], 0, 1 << 26), "push_kernel", (1 << 30) // (KBLOCK * 32), 2 << 30)
    del storage
    narrow = big.view(1, 1 << 28, 1)                       # the same GiB as 256 Mi rows of 4 bytes
    idx = torch.randperm(1 << 28, device=DEV)[: 1 << 23].contiguous()
    run("random_rows_4B", lambda: ops.gather_rows([narrow], idx, 1, 1 << 28), "gather_kernel", blocks_plain(1 << 23, 4, 4), (1 << 23) * (8 + 8))
    rec = big.view(1, 1 << 25, 8)                          # 32 Mi rows of 32 bytes
    idx32 = torch.randperm(1 << 25, device=DEV)[: 1 << 23].contiguous()
    run("random_rows_32B", lambda: ops.gather_rows([rec], idx32, 1, 1 << 25), "gather_kernel", blocks_plain(1 << 23, 32, 16), (1 << 23) * (64 + 8))
    wide = torch.empty(1, 1 << 23, 48, device=DEV).normal_()  # 8 Mi rows of 192 bytes = 1.5 GiB
    idxw = torch.randperm(1 << 23, device=DEV)[: 1 << 21].contiguous()
    run("random_rows_192B", lambda: ops.gather_rows([wide], idxw, 1, 1 << 23), "gather_kernel", blocks_plain(1 << 21, 192, 16), (1 << 21) * (384 + 8))
    del big, wide, narrow, rec
    torch.cuda.empty_cache()

    # ---- product launches at config 2
    T, N, obs, act = 24, 4096, 48, 12
    S, B = T * N, T * N // 4
    flag = lambda p: torch.rand(T, N, 1, device=DEV) < p  # noqa: E731
    leaves = {"observation": f(T, N, obs), "mean": f(T, N, act), "std": f(T, N, act), "action": f(T, N, act), "logp": f(T, N, 1),
              "value": f(T, N, 1), "next_observation": f(T, N, obs), "reward": f(T, N, 1), "terminated": flag(0.01),
              "truncated": flag(0.005), "done": flag(0.015), "next_value": f(T, N, 1), "advantage": f(T, N, 1), "return": f(T, N, 1)}
    narrow_names = [k for k, v in leaves.items() if ops.RecordPack.eligible(v)]
    pack = ops.RecordPack({k: leaves[k] for k in narrow_names})
    run("pack_rows", pack.build, "pack_rows_kernel", -(-S // KBLOCK), S * (pack.used_bytes + pack.record_bytes))
    perm = torch.randperm(S, device=DEV)[:B].contiguous()
    hot_plain, hot_packed = ["observation", "action"], ["logp", "value", "advantage", "return", "done"]
    wide_blocks = lambda names: sum(blocks_plain(B, leaves[k][0, 0].numel() * 4, 16) for k in names)  # noqa: E731
    record_blocks = lambda p: -(-(B * -(-p.used_bytes // 16)) // (KBLOCK * ITEMS))  # record-major: 16-byte chunks  # noqa: E731
    hot_bytes = B * (2 * (4 * obs + 4 * act + 17) + 8)
    run("gather_minibatch_hot_leaves", lambda: ops.gather_rows_packed([leaves[k] for k in hot_plain], pack, hot_packed, perm, T, N),
        "gather_kernel", wide_blocks(hot_plain) + record_blocks(pack), hot_bytes)
    hot_pack = ops.RecordPack({k: leaves[k] for k in ("observation", "action", "logp", "advantage", "return", "done")})
    run("pack_hot_record", hot_pack.build, "gather_kernel", blocks_plain(S, 192, 16) + blocks_plain(S, 48, 16),
        S * (hot_pack.used_bytes + hot_pack.record_bytes), {"gather_kernel": 1, "pack_rows_kernel": 1})  # wide copy + narrow entries
    run("gather_minibatch_hot_record", lambda: ops.gather_rows_packed([], hot_pack, list(hot_pack.leaves), perm, T, N), "gather_kernel",
        record_blocks(hot_pack), B * (2 * hot_pack.used_bytes + 8))
    all_plain = [k for k in leaves if k not in narrow_names]
    all_bytes = B * (2 * 555 + 8)
    run("gather_minibatch_all_leaves", lambda: ops.gather_rows_packed([leaves[k] for k in all_plain], pack, narrow_names, perm, T, N),
        "gather_kernel", wide_blocks(all_plain) + record_blocks(pack), all_bytes)
    plain_blocks = wide_blocks(all_plain) + 6 * blocks_plain(B, 4, 4) + 3 * (-(-((B + 3) // 4) // KBLOCK))
    run("gather_minibatch_all_plain", lambda: ops.gather_rows(list(leaves.values()), perm, T, N), "gather_kernel", plain_blocks, all_bytes)
    hot6 = ["observation", "action", "logp", "advantage", "return", "done"]
    hot6_blocks = (wide_blocks(["observation", "action"]) + 3 * blocks_plain(B, 4, 4) + (-(-((B + 3) // 4) // KBLOCK)))
    run("gather_minibatch_hot_plain", lambda: ops.gather_rows([leaves[k] for k in hot6], perm, T, N), "gather_kernel", hot6_blocks,
        B * (2 * 253 + 8))
    del leaves, pack, hot_pack
    torch.cuda.empty_cache()

    # ---- the fused objective at roofline scale
    Bl, A = (1 << 20) * 24 // 4, 12
    a = dict(advantage=f(Bl, 1), old_logp=f(Bl, 1) - 12, action=f(Bl, A), mean=f(Bl, A), std=torch.rand(Bl, A, device=DEV) + 0.5,
             ret=f(Bl, 1), curr_value=f(Bl, 1), old_value=f(Bl, 1))
    kw = dict(clip=0.2, value_clip=None, w_sur=1.0, w_val=0.5, w_ent=0.01)
    loss_blocks = -(-Bl // 252)  # 21 rows x 4 waves x 3 rounds per block at A = 12
    matrix_bytes = Bl * (8 + 3 * 4 * A + 8 + 2 * 4 * A + 4 + 16)
    v = dict(a, std=torch.rand(A, device=DEV) + 0.5)
    run("loss_std_vector_1m", lambda: ops.ppo_loss_fwd_bwd(*v.values(), **kw), "ppo_loss_rowgroup_kernel", loss_blocks,
        matrix_bytes - Bl * 8 * A)
    run("loss_std_matrix_1m", lambda: ops.ppo_loss_fwd_bwd(*a.values(), **kw), "ppo_loss_rowgroup_kernel", loss_blocks, matrix_bytes)
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    Path(out_dir, "cases.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
