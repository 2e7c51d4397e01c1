"""CPU oracle for the rollout + PPO-update hot path — TEST INFRASTRUCTURE ONLY.

`oracle/cusrl_oracle.c` restates the reference's algorithm (chengruiz/cusrl, file:line cited
per function in the C source); this module is its numpy/ctypes face.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg — never by
``cusrl_amd``.  Parity status: PINNED — checked against golden vectors produced by running
the reference itself (``tests/golden/make_golden.py``) and against the reference's own
known-answer tests (``tests/test_oracle_golden.py``).
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libcusrl_oracle.so"
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_u8 = ctypes.POINTER(ctypes.c_uint8)
c_i64 = ctypes.POINTER(ctypes.c_int64)
c_u32 = ctypes.POINTER(ctypes.c_uint32)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (seconds)."""
    src = _HERE / "cusrl_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libcusrl_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.oracle_next_value.restype = ctypes.c_int64
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _u8(x):
    return np.ascontiguousarray(np.asarray(x).astype(np.uint8))


def _p(arr, typ):
    return None if arr is None else arr.ctypes.data_as(typ)


# ------------------------------------------------------------------------------------------ a1
def buffer_push(step: np.ndarray, storage: np.ndarray, cursor: int) -> None:
    step = np.ascontiguousarray(step)
    assert storage.flags.c_contiguous and storage.dtype == step.dtype and storage.shape[1:] == step.shape
    lib().oracle_buffer_push(
        step.ctypes.data_as(ctypes.c_void_p), storage.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(cursor), ctypes.c_int64(step.nbytes),
    )


# ------------------------------------------------------------------------------------------ a3
def next_value(value, terminated, truncated, last_value, trunc_values=None, bootstrap=True, termination_value=0.0):
    value = _f32(value)
    T, N, D = value.shape
    out = np.empty_like(value)
    tv = None if trunc_values is None else _f32(trunc_values)
    term, trunc, last = _u8(terminated), _u8(truncated), _f32(last_value)
    k = lib().oracle_next_value(
        _p(value, c_f), _p(term, c_u8), _p(trunc, c_u8), _p(last, c_f), _p(tv, c_f),
        ctypes.c_int(int(bootstrap)), ctypes.c_float(termination_value), _p(out, c_f),
        ctypes.c_int64(T), ctypes.c_int64(N), ctypes.c_int64(D),
    )
    return out, int(k)


# ------------------------------------------------------------------------------------------ a4
def gae(reward, done, value, next_value_, gamma, lamda, lamda_value=None):
    reward, value, next_value_ = _f32(reward), _f32(value), _f32(next_value_)
    T, N, D = reward.shape
    done = _u8(done)
    assert done.size == T * N
    adv, ret = np.empty_like(reward), np.empty_like(reward)
    lib().oracle_gae(
        _p(reward, c_f), _p(done, c_u8), _p(value, c_f), _p(next_value_, c_f),
        ctypes.c_double(gamma), ctypes.c_double(lamda), ctypes.c_double(-1.0 if lamda_value is None else lamda_value),
        _p(adv, c_f), _p(ret, c_f), ctypes.c_int64(T), ctypes.c_int64(N), ctypes.c_int64(D),
    )
    return adv, ret


# ------------------------------------------------------------------------------------------ a5
def var_mean(x):
    x = _f32(x)
    D = x.shape[-1]
    rows = x.size // D
    mean, var = np.empty(D, np.float32), np.empty(D, np.float32)
    lib().oracle_var_mean(_p(x, c_f), ctypes.c_int64(rows), ctypes.c_int64(D), _p(mean, c_f), _p(var, c_f))
    return var, mean


def normalize(x, mean, var):
    out = _f32(x).copy()
    D = out.shape[-1]
    mean, var = _f32(mean), _f32(var)
    lib().oracle_normalize(_p(out, c_f), ctypes.c_int64(out.size // D), ctypes.c_int64(D), _p(mean, c_f), _p(var, c_f))
    return out


# ------------------------------------------------------------------------------------------ a6
def merge_mean_var(means, vars_):
    means, vars_ = _f32(means), _f32(vars_)
    W, D = means.shape
    mean, var = np.empty(D, np.float32), np.empty(D, np.float32)
    lib().oracle_merge_mean_var(_p(means, c_f), _p(vars_, c_f), ctypes.c_int64(W), ctypes.c_int64(D), _p(mean, c_f), _p(var, c_f))
    return mean, var


# ------------------------------------------------------------------------------------------ policy statistics
def policy_stats(old_mean, old_std, new_mean, new_std, action, old_logp, advantage):
    """cusrl/hook/on_policy/stats.py:28-40 for a Normal policy: (mean KL(old || new) summed over action dims,
    mean advantage * exp(logp_new(action) - old_logp), mean new_std), computed in float64."""
    mp, sp, mq, sq, x = (np.asarray(a, np.float64) for a in (old_mean, old_std, new_mean, new_std, action))
    var_ratio = (sp / sq) ** 2  # torch.distributions.kl._kl_normal_normal
    kl = (0.5 * (var_ratio + ((mp - mq) / sq) ** 2 - 1.0 - np.log(var_ratio))).sum(-1)
    logp = (-((x - mq) ** 2) / (2.0 * sq**2) - np.log(sq) - 0.5 * np.log(2.0 * np.pi)).sum(-1)
    weight = np.exp(logp - np.asarray(old_logp, np.float64).reshape(-1))
    advantage = np.asarray(advantage, np.float64).reshape(weight.size, -1)
    return float(kl.mean()), float((advantage * weight[:, None]).mean()), float(sq.mean())


# ------------------------------------------------------------------------------------------ gradient clipping
def clip_grad_norm(grad, max_norm):
    """cusrl/hook/on_policy/gradient_clipping.py:67-83 -> torch.nn.utils.clip_grad_norm_ (norm_type 2):
    ``total = ||g||_2``; ``g * min(max_norm / (total + 1e-6), 1)`` in fp32.  Returns (clipped copy, total);
    ``max_norm=None`` only measures."""
    grad = _f32(grad)
    total = np.float32(np.sqrt(np.sum(grad.astype(np.float64) ** 2)))
    if max_norm is None:
        return grad.copy(), total
    coef = np.float32(max_norm) / (total + np.float32(1e-6))
    return grad * np.minimum(coef, np.float32(1.0)), total


def adam_step(param, grad, exp_avg, exp_avg_sq, step, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              decoupled=False, grad_scale=1.0):
    """One step of torch.optim.Adam / AdamW (the optimizer cusrl/preset/ppo.py builds), fp32 like torch's kernels:
    returns (param, exp_avg, exp_avg_sq, step + 1).  ``grad_scale`` is the clipping coefficient applied first."""
    f = np.float32
    param, grad, exp_avg, exp_avg_sq = (_f32(a).copy() for a in (param, grad, exp_avg, exp_avg_sq))
    beta1, beta2 = float(betas[0]), float(betas[1])  # python floats (doubles): 1 - beta is rounded to fp32 afterwards
    step = step + 1
    grad = grad * f(grad_scale)
    if weight_decay:
        if decoupled:
            param = param * (f(1.0) - f(lr) * f(weight_decay))
        else:
            grad = grad + f(weight_decay) * param
    exp_avg = exp_avg + (grad - exp_avg) * f(1.0 - beta1)
    exp_avg_sq = f(beta2) * exp_avg_sq + f(1.0 - beta2) * (grad * grad)
    step_size = f(lr / (1.0 - beta1**step))
    denom = np.sqrt(exp_avg_sq) / f(np.sqrt(1.0 - beta2**step)) + f(eps)
    param = param - step_size * (exp_avg / denom)
    return param.astype(f), exp_avg.astype(f), exp_avg_sq.astype(f), step


# ------------------------------------------------------------------------------------------ a7/a8
def gather_rows(storage: np.ndarray, indices, temporal: bool = False) -> np.ndarray:
    storage = np.ascontiguousarray(storage)
    T, N = storage.shape[:2]
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    B = indices.size
    row_bytes = storage[0, 0].nbytes
    shape = ((T, B) if temporal else (B,)) + storage.shape[2:]
    out = np.empty(shape, storage.dtype)
    lib().oracle_gather_rows(
        storage.ctypes.data_as(ctypes.c_void_p), _p(indices, c_i64), out.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(B), ctypes.c_int64(T), ctypes.c_int64(N), ctypes.c_int64(row_bytes), ctypes.c_int(int(temporal)),
    )
    return out


def window_slots(start, env, sequence_len: int, capacity: int, parallelism: int, cursor: int | None) -> np.ndarray:
    """cusrl/sampler/random_sampler.py:101-113 restated: physical flat slots ``[sequence_len, B]`` of windows that start at
    logical step ``start[b]`` in env ``env[b]``; ``cursor`` = oldest row of a full ring (None while it is filling)."""
    start, env = np.asarray(start, np.int64), np.asarray(env, np.int64)
    time = start[None, :] + np.arange(sequence_len, dtype=np.int64)[:, None]
    if cursor is not None:
        time = (cursor + time) % capacity
    return time * parallelism + env[None, :]


def sequence_lengths(done: np.ndarray) -> np.ndarray:
    """cusrl/nn/utils/recurrent.py:63-92 restated: lengths of the done-split sequences, env-major order."""
    close = np.array(done).reshape(done.shape[0], -1).astype(bool)
    close[-1] = True  # the last step always closes a sequence
    ends = np.nonzero(close.T.reshape(-1))[0]
    return np.diff(np.concatenate(([-1], ends))).astype(np.int64)


def sequence_layout(done: np.ndarray) -> tuple[np.ndarray, int]:
    """Where ``split_and_pad_sequences`` (recurrent.py:215-252) puts every slot: ``dest[t * N + n] = pos * Ns + seq`` with
    ``seq`` the env-major index of the slot's sequence and ``pos`` its step inside that sequence; also returns ``Ns``."""
    close = np.array(done).reshape(done.shape[0], -1).astype(bool)
    L, N = close.shape
    close[-1] = True
    counts = close.sum(axis=0)
    first = np.concatenate(([0], np.cumsum(counts)[:-1]))
    before = np.cumsum(close, axis=0) - close                     # sequences of this env closed before step t
    restart = np.where(close, np.arange(1, L + 1)[:, None], 0)    # a close at t makes t + 1 the next sequence's start
    start = np.maximum.accumulate(np.vstack((np.zeros((1, N), np.int64), restart[:-1])), axis=0)
    pos = np.arange(L)[:, None] - start
    num_sequences = int(counts.sum())
    return ((pos * num_sequences) + first[None, :] + before).reshape(-1).astype(np.int64), num_sequences


def gather_memory(memory: np.ndarray, done: np.ndarray) -> np.ndarray:
    """cusrl/nn/utils/recurrent.py:124-157 restated: the state of the sequence still open at the end of each env's
    column (sequences are numbered env-major, a done before the last step opens a new one), cleared where done[-1]."""
    done = np.asarray(done).reshape(done.shape[0], -1).astype(bool)
    last = np.cumsum(done[:-1].sum(axis=0)) + np.arange(done.shape[1])
    out = np.array(memory[last], copy=True)
    out[done[-1]] = 0
    return out


class Mt19937:
    """torch's CPU generator stream as consumed by ``torch.randperm`` (see the C source)."""

    def __init__(self, seed: int):
        self.state = np.zeros(625, np.uint32)
        lib().oracle_mt19937_seed(_p(self.state, c_u32), ctypes.c_uint64(seed))

    def randperm(self, n: int) -> np.ndarray:
        out = np.empty(n, np.int64)
        lib().oracle_randperm(_p(self.state, c_u32), ctypes.c_int64(n), _p(out, c_i64))
        return out


def mini_batch_indices(seed_or_gen, num_samples, num_epochs, num_mini_batches, shuffle=True):
    """cusrl/sampler/mini_batch_sampler.py:52-78 — the index vector of every minibatch, in order."""
    gen = seed_or_gen if isinstance(seed_or_gen, Mt19937) else Mt19937(seed_or_gen)
    epoch_indices = gen.randperm(num_samples)
    result = []
    for epoch in range(num_epochs):
        mbs = num_mini_batches if isinstance(num_mini_batches, int) else num_mini_batches[epoch]
        size = num_samples // mbs
        if shuffle and epoch > 0:
            epoch_indices = gen.randperm(num_samples)
        result.append([epoch_indices[j * size:(j + 1) * size].copy() for j in range(mbs)])
    return result


# ------------------------------------------------------------------------------------------ a9-a13
def normal_logp_entropy(action, mean, std):
    action, mean, std = _f32(action), _f32(mean), _f32(std)
    B, A = mean.shape
    logp, ent = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
    lib().oracle_normal_logp_entropy(_p(action, c_f), _p(mean, c_f), _p(std, c_f), _p(logp, c_f), _p(ent, c_f), ctypes.c_int64(B), ctypes.c_int64(A))
    return logp, ent


def normal_kl(mean_p, std_p, mean_q, std_q):
    mean_p, std_p, mean_q, std_q = _f32(mean_p), _f32(std_p), _f32(mean_q), _f32(std_q)
    B, A = mean_p.shape
    kl = np.empty((B, 1), np.float32)
    lib().oracle_normal_kl(_p(mean_p, c_f), _p(std_p, c_f), _p(mean_q, c_f), _p(std_q, c_f), _p(kl, c_f), ctypes.c_int64(B), ctypes.c_int64(A))
    return kl


def ppo_loss(advantage, old_logp, action, mean, std, ret, curr_value, old_value=None, *, clip=0.2,
             value_clip=None, w_sur=1.0, w_val=0.5, w_ent=0.01):
    """Returns dict(losses[3]=(value, surrogate, entropy), logp, entropy, ratio, d_mean, d_std, d_value)."""
    advantage, old_logp, action = _f32(advantage), _f32(old_logp), _f32(action)
    mean, std, ret, curr_value = _f32(mean), _f32(std), _f32(ret), _f32(curr_value)
    old_value = None if old_value is None else _f32(old_value)
    B, A = mean.shape
    D = ret.shape[-1]
    out = dict(
        losses=np.empty(3, np.float32), logp=np.empty((B, 1), np.float32), entropy=np.empty((B, 1), np.float32),
        ratio=np.empty((B, 1), np.float32), d_mean=np.empty((B, A), np.float32), d_std=np.empty((B, A), np.float32),
        d_value=np.empty((B, D), np.float32),
    )
    lib().oracle_ppo_loss(
        _p(advantage, c_f), _p(old_logp, c_f), _p(action, c_f), _p(mean, c_f), _p(std, c_f), _p(ret, c_f),
        _p(curr_value, c_f), _p(old_value, c_f), ctypes.c_int64(B), ctypes.c_int64(A), ctypes.c_int64(D),
        ctypes.c_double(clip), ctypes.c_double(-1.0 if value_clip is None else value_clip),
        ctypes.c_double(w_sur), ctypes.c_double(w_val), ctypes.c_double(w_ent),
        _p(out["losses"], c_f), _p(out["logp"], c_f), _p(out["entropy"], c_f), _p(out["ratio"], c_f),
        _p(out["d_mean"], c_f), _p(out["d_std"], c_f), _p(out["d_value"], c_f),
    )
    return out


def ppo_loss_f64(advantage, old_logp, action, mean, std, ret, curr_value, old_value=None, *, clip=0.2, value_clip=None,
                 w_sur=1.0, w_val=0.5, w_ent=0.01, flip_clip_side=None):
    """:func:`ppo_loss` evaluated in float64 from the same float32 inputs — the yardstick for GRADIENT parity.

    Why a second restatement: the fp32 forms (the reference's torch ops, ``oracle_ppo_loss``, the HIP kernel) each round
    ``logp`` (a sum of A terms of magnitude ~|logp|) in their own order, and ``ratio = exp(logp - old_logp)`` turns that
    absolute noise (~6e-8 * |logp|) into a RELATIVE error of every gradient element of the row — 1e-6..1e-5 at |logp| ~ 15-60
    — before any element-wise cancellation.  Two fp32 evaluations therefore cannot be held to 1e-5 against each other, but each
    can be held to the float64 value.  Same formulas and citations as :func:`ppo_loss` (distribution.py:207-213, common.py:35-41,
    ppo.py:10-18, value.py:85-89,121-137); ``std`` may be the [A] vector (then ``d_std`` is its [A] gradient).
    ``flip_clip_side``: bool [B] — rows whose ratio is to be treated as lying on the OTHER side of the clip bound it is
    nearest to (a ratio within fp32 noise of ``1 +- clip`` may legitimately fall on either side; tests accept both)."""
    f = lambda x: np.asarray(x, np.float64)  # noqa: E731
    adv, old_logp = f(advantage).reshape(-1), f(old_logp).reshape(-1)
    x, mu, sg, R, cv = f(action), f(mean), f(std), f(ret), f(curr_value)
    B, A = mu.shape
    D = R.shape[-1]
    vector = sg.ndim == 1
    sgb = np.broadcast_to(sg, mu.shape)
    diff = x - mu
    logp = (-(diff * diff) / (2.0 * sgb * sgb) - np.log(sgb) - np.log(np.sqrt(2.0 * np.pi))).sum(-1)
    entropy = (0.5 + 0.5 * np.log(2.0 * np.pi) + np.log(sgb)).sum(-1)
    ratio = np.exp(logp - old_logp)
    lo, hi = np.float64(np.float32(1.0 - clip)), np.float64(np.float32(1.0 + clip))
    inside = (ratio >= lo) & (ratio <= hi)
    clipped = np.clip(ratio, lo, hi)
    if flip_clip_side is not None:
        flip = np.asarray(flip_clip_side, bool).reshape(-1)
        nearest = np.where(np.abs(ratio - lo) < np.abs(ratio - hi), lo, hi)
        clipped = np.where(flip, np.where(inside, nearest, ratio), clipped)
        inside = inside ^ flip
    s1, s2 = adv * ratio, adv * clipped
    d_ratio = np.where(s1 < s2, adv, np.where(s1 > s2, np.where(inside, adv, 0.0), 0.5 * adv + np.where(inside, 0.5 * adv, 0.0)))
    dlp = (-w_sur / B) * d_ratio * ratio
    var = sgb * sgb
    d_mean = dlp[:, None] * (diff / var)
    d_std = dlp[:, None] * ((diff * diff) / (var * sgb) - 1.0 / sgb) + (-w_ent / B) / sgb
    e1 = cv - R
    if value_clip is None:
        value_loss, g = (e1 * e1).mean() * w_val, 2.0 * e1
    else:
        v = f(old_value)
        dv = cv - v
        e2 = (v + np.clip(dv, -value_clip, value_clip)) - R
        l1, l2 = e1 * e1, e2 * e2
        g2 = np.where((dv >= -value_clip) & (dv <= value_clip), 2.0 * e2, 0.0)
        value_loss = np.maximum(l1, l2).mean() * w_val
        g = np.where(l1 > l2, 2.0 * e1, np.where(l1 < l2, g2, 0.5 * (2.0 * e1 + g2)))
    return dict(
        losses=np.array([value_loss, -np.minimum(s1, s2).mean() * w_sur, -entropy.mean() * w_ent]),
        logp=logp[:, None], entropy=entropy[:, None], ratio=ratio[:, None], d_mean=d_mean,
        d_std=d_std.sum(0) if vector else d_std, d_value=(w_val / (B * D)) * g,
        clip_margin=np.minimum(np.abs(ratio - lo), np.abs(ratio - hi)),
    )


def policy_terms_f64(mean, std, action, old_logp, g_logp=None, g_entropy=None, g_logp_ratio=None, g_ratio=None):
    """What OnPolicyPreparation.objective leaves in the batch (cusrl/hook/on_policy/common.py:29-43) for a Normal policy —
    ``logp`` / ``entropy`` as sums over the action dims of torch.distributions.Normal.log_prob / entropy
    (cusrl/nn/module/distribution.py:207-213), ``logp_ratio = logp - old_logp``, ``ratio = exp(logp_ratio)`` — evaluated in
    float64, and (given gradients wrt those four [B] outputs, None = no gradient) the vector-Jacobian products wrt ``mean`` and
    ``std`` that autograd forms through the same ops.  ``std`` may be the [A] vector (``d_std`` is then its [A] gradient)."""
    f = lambda x: np.asarray(x, np.float64)  # noqa: E731
    x, mu, sg, old = f(action), f(mean), f(std), f(old_logp).reshape(-1)
    vector = sg.ndim == 1
    sgb = np.broadcast_to(sg, mu.shape)
    diff = x - mu
    logp = (-(diff * diff) / (2.0 * sgb * sgb) - np.log(sgb) - np.log(np.sqrt(2.0 * np.pi))).sum(-1)
    entropy = (0.5 + 0.5 * np.log(2.0 * np.pi) + np.log(sgb)).sum(-1)
    lr = logp - old
    ratio = np.exp(lr)
    out = dict(logp=logp[:, None], entropy=entropy[:, None], logp_ratio=lr[:, None], ratio=ratio[:, None])
    if g_logp is None and g_entropy is None and g_logp_ratio is None and g_ratio is None:
        return out
    zero = np.zeros_like(logp)
    g = lambda v: zero if v is None else f(v).reshape(-1)  # noqa: E731
    G = g(g_logp) + g(g_logp_ratio) + g(g_ratio) * ratio
    var = sgb * sgb
    out["d_mean"] = G[:, None] * (diff / var)
    d_std = G[:, None] * ((diff * diff) / (var * sgb) - 1.0 / sgb) + g(g_entropy)[:, None] / sgb
    out["d_std"] = d_std.sum(0) if vector else d_std
    return out


def categorical_terms_f64(logits, action, old_logp, g_logp=None, g_entropy=None, g_logp_ratio=None, g_ratio=None):
    """The same four terms for a one-hot categorical policy (distribution.py:354-362: OneHotCategorical.log_prob / entropy;
    ``log p`` clamped to the smallest finite float32 like torch.distributions.Categorical.entropy) and their VJP wrt logits."""
    f = lambda x: np.asarray(x, np.float64)  # noqa: E731
    z, onehot, old = f(logits), f(action), f(old_logp).reshape(-1)
    taken = onehot.argmax(-1)
    norm = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)) + z.max(-1, keepdims=True)
    lp = np.maximum(z - norm, np.float64(np.finfo(np.float32).min))
    p = np.exp(lp)
    entropy = -(p * lp).sum(-1)
    rows = np.arange(z.shape[0])
    logp = (z - norm)[rows, taken]
    lr = logp - old
    ratio = np.exp(lr)
    out = dict(logp=logp[:, None], entropy=entropy[:, None], logp_ratio=lr[:, None], ratio=ratio[:, None])
    if g_logp is None and g_entropy is None and g_logp_ratio is None and g_ratio is None:
        return out
    zero = np.zeros_like(logp)
    g = lambda v: zero if v is None else f(v).reshape(-1)  # noqa: E731
    G = g(g_logp) + g(g_logp_ratio) + g(g_ratio) * ratio
    hot = np.zeros_like(z)
    hot[rows, taken] = 1.0
    out["d_logits"] = G[:, None] * (hot - p) - g(g_entropy)[:, None] * p * (lp + entropy[:, None])
    return out


def reward_shaping(reward, scale=1.0, shift=0.0, lower=None, upper=None):
    """RewardShaping.post_step (cusrl/hook/mdp/reward.py:43-47): ``reward.mul_(scale).add_(shift)`` — product and sum rounded
    separately in fp32 — then ``clamp_(min, max)``; returns the new array."""
    out = np.asarray(reward, np.float32) * np.float32(scale)
    out = (out + np.float32(shift)).astype(np.float32)
    if lower is not None:
        out = np.where(out < np.float32(lower), np.float32(lower), out)
    if upper is not None:
        out = np.where(out > np.float32(upper), np.float32(upper), out)
    return out.astype(np.float32)


def running_mean_std_update(mean, var, count, batch, epsilon=1e-8, max_count=None):
    """One ``RunningMeanStd.update(batch)`` (cusrl/nn/layer/rms.py:140-167 with cusrl/nn/utils/normalization.py:15-50,80-93):
    population statistics of the batch rows, merged with weights count : rows; fp32 like the reference's tensors.
    Returns ``(mean, var, std, count)``."""
    batch = np.asarray(batch, np.float32).reshape(-1, np.shape(batch)[-1])
    n = batch.shape[0]
    if n == 0:
        return mean, var, np.sqrt(var + np.float32(epsilon)), count
    b64 = batch.astype(np.float64)
    batch_mean = b64.mean(0).astype(np.float32)
    batch_var = b64.var(0).astype(np.float32)  # torch.var_mean(correction=0)
    w_sum = count + n
    w_old, w_new = np.float32(count / w_sum), np.float32(n / w_sum)
    delta = batch_mean - np.asarray(mean, np.float32)
    new_mean = (mean + delta * w_new).astype(np.float32)
    new_var = (var + ((batch_var - var) * w_new + delta * delta * np.float32((count / w_sum) * (n / w_sum)))).astype(np.float32)
    total = count + n
    if max_count is not None and total > max_count:
        total = max_count
    return new_mean, new_var, np.sqrt(new_var + np.float32(epsilon)).astype(np.float32), total


def amp_prepare(state, next_state, columns, dataset, picks, mean, var, count, clamp=10.0, epsilon=1e-8, max_count=None):
    """AdversarialMotionPrior.post_step up to the discriminator (cusrl/hook/auxiliary/amp.py:112-128):
    ``agent = cat(state[:, columns], next_state[:, columns])``, ``expert = dataset[picks]``, ``rms.update(agent)``,
    ``rms.update(expert)``, both normalised (``(x - mean) / std`` clamped to +-clamp, rms.py:198-203) with the statistics
    after both updates.  Returns ``(agent, expert, mean, var, std, count)``."""
    agent = np.concatenate([np.asarray(state, np.float32)[:, columns], np.asarray(next_state, np.float32)[:, columns]], -1)
    expert = np.asarray(dataset, np.float32)[np.asarray(picks)]
    mean, var, std, count = running_mean_std_update(np.asarray(mean, np.float32), np.asarray(var, np.float32), count, agent,
                                                    epsilon, max_count)
    mean, var, std, count = running_mean_std_update(mean, var, count, expert, epsilon, max_count)

    def normalise(x):
        out = ((x - mean) / std).astype(np.float32)
        return out if clamp is None else np.clip(out, -np.float32(clamp), np.float32(clamp))

    return normalise(agent), normalise(expert), mean, var, std, count


def amp_style_reward(logit, scale):
    """``reward_scale * -log(clamp(1 - 1 / (1 + exp(-logit)), min=1e-4))`` (amp.py:131) in fp32, the reference's op order."""
    logit = np.asarray(logit, np.float32)
    p = (np.float32(1) - np.float32(1) / (np.float32(1) + np.exp(-logit))).astype(np.float32)
    return (np.float32(scale) * -np.log(np.maximum(p, np.float32(1e-4)))).astype(np.float32)


def mse_loss(prediction, target):
    """``nn.MSELoss()(prediction, target)`` (rnd.py:80) and its gradient wrt ``prediction`` in float64."""
    p, t = np.asarray(prediction, np.float64), np.asarray(target, np.float64)
    d = p - t
    return float((d * d).mean()), 2.0 * d / d.size


def gradient_error(candidate, reference) -> float:
    """max |candidate - reference| / max |reference|: the error of a gradient TENSOR in units of its largest entry (what
    the optimizer step sees), robust against elements that cancel to ~0 where an element-wise relative error is meaningless."""
    reference = np.asarray(reference, np.float64)
    scale = np.abs(reference).max()
    return float(np.abs(np.asarray(candidate, np.float64) - reference).max() / (scale if scale > 0 else 1.0))


def gru_sequence(x, h0, weights, lengths=None):
    """``torch.nn.GRU`` over a time-major batch, restated in numpy float64 (the recurrent backbone the reference wraps:
    cusrl/nn/module/rnn.py:21-120, ``nn.GRU``; the cell itself is PyTorch's — r, z = sigmoid(W_i x + b_i + W_h h + b_h),
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) n + z h — pinned against outputs and memories recorded
    from the reference's own Rnn wrapper, tests/golden/recurrent.npz).  ``weights``: per layer (w_ih [3H, I], w_hh [3H, H],
    b_ih [3H] | None, b_hh [3H] | None); ``h0`` [layers, B, H]; ``lengths`` [B]: the packed-sequence result (state frozen and
    zero output from each sequence's end on).  Returns (output [L, B, H], h_n [layers, B, H])."""
    x = np.asarray(x, np.float64)
    L, B, _ = x.shape
    finals = []
    for layer, (w_ih, w_hh, b_ih, b_hh) in enumerate(weights):
        w_ih, w_hh = np.asarray(w_ih, np.float64), np.asarray(w_hh, np.float64)
        H = w_hh.shape[1]
        b_ih = np.zeros(3 * H) if b_ih is None else np.asarray(b_ih, np.float64)
        b_hh = np.zeros(3 * H) if b_hh is None else np.asarray(b_hh, np.float64)
        h = np.zeros((B, H)) if h0 is None else np.asarray(h0[layer], np.float64).copy()
        out = np.zeros((L, B, H))
        for t in range(L):
            gi, gh = x[t] @ w_ih.T + b_ih, h @ w_hh.T + b_hh
            r = 1.0 / (1.0 + np.exp(-(gi[:, :H] + gh[:, :H])))
            z = 1.0 / (1.0 + np.exp(-(gi[:, H:2 * H] + gh[:, H:2 * H])))
            n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            nxt = (1.0 - z) * n + z * h
            live = np.ones(B, bool) if lengths is None else t < np.asarray(lengths)
            h = np.where(live[:, None], nxt, h)
            out[t] = np.where(live[:, None], nxt, 0.0)
        finals.append(h)
        x = out
    return x.astype(np.float32), np.stack(finals).astype(np.float32)


def rnn_sequence(x, h0, weights, lengths=None, relu=False):
    """``torch.nn.RNN`` (h' = tanh | relu(W_ih x + b_ih + W_hh h + b_hh)) in numpy float64, arguments as :func:`gru_sequence`;
    pinned against the reference's recorded packed-sequence run (tests/golden/recurrent_packed.npz, rnn_* entries)."""
    x = np.asarray(x, np.float64)
    L, B, _ = x.shape
    finals = []
    for layer, (w_ih, w_hh, b_ih, b_hh) in enumerate(weights):
        w_ih, w_hh = np.asarray(w_ih, np.float64), np.asarray(w_hh, np.float64)
        H = w_hh.shape[1]
        bias = (0.0 if b_ih is None else np.asarray(b_ih, np.float64)) + (0.0 if b_hh is None else np.asarray(b_hh, np.float64))
        h = np.zeros((B, H)) if h0 is None else np.asarray(h0[layer], np.float64).copy()
        out = np.zeros((L, B, H))
        for t in range(L):
            pre = x[t] @ w_ih.T + h @ w_hh.T + bias
            nxt = np.maximum(pre, 0.0) if relu else np.tanh(pre)
            live = (np.ones(B, bool) if lengths is None else t < np.asarray(lengths))[:, None]
            h = np.where(live, nxt, h)
            out[t] = np.where(live, nxt, 0.0)
        finals.append(h)
        x = out
    return x.astype(np.float32), np.stack(finals).astype(np.float32)


def lstm_sequence(x, state, weights, lengths=None):
    """``torch.nn.LSTM`` over a time-major batch in numpy float64 (gate order i, f, g, o; c' = f c + i g, h' = o tanh(c')) —
    the default recurrent core of the reference's preset (cusrl/preset/ppo.py:189, wrapped by nn/module/rnn.py:21-120);
    pinned like :func:`gru_sequence` against tests/golden/recurrent.npz.  ``state`` = (h0, c0) [layers, B, H] or None.
    Returns (output [L, B, H], (h_n, c_n))."""
    x = np.asarray(x, np.float64)
    L, B, _ = x.shape
    last_h, last_c = [], []
    sigmoid = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    for layer, (w_ih, w_hh, b_ih, b_hh) in enumerate(weights):
        w_ih, w_hh = np.asarray(w_ih, np.float64), np.asarray(w_hh, np.float64)
        H = w_hh.shape[1]
        bias = (0.0 if b_ih is None else np.asarray(b_ih, np.float64)) + (0.0 if b_hh is None else np.asarray(b_hh, np.float64))
        h = np.zeros((B, H)) if state is None else np.asarray(state[0][layer], np.float64).copy()
        c = np.zeros((B, H)) if state is None else np.asarray(state[1][layer], np.float64).copy()
        out = np.zeros((L, B, H))
        for t in range(L):
            pre = x[t] @ w_ih.T + h @ w_hh.T + bias
            i, f, g, o = sigmoid(pre[:, :H]), sigmoid(pre[:, H:2 * H]), np.tanh(pre[:, 2 * H:3 * H]), sigmoid(pre[:, 3 * H:])
            c_next = f * c + i * g
            h_next = o * np.tanh(c_next)
            live = (np.ones(B, bool) if lengths is None else t < np.asarray(lengths))[:, None]
            h, c = np.where(live, h_next, h), np.where(live, c_next, c)
            out[t] = np.where(live, h_next, 0.0)
        last_h.append(h), last_c.append(c)
        x = out
    return x.astype(np.float32), (np.stack(last_h).astype(np.float32), np.stack(last_c).astype(np.float32))


def categorical_sample(logits, noise):
    """Acting side of a one-hot categorical policy, restated in numpy float64: cusrl/nn/module/distribution.py:332-366
    (``OneHotCategorical(logits).sample()`` and ``log_prob`` of the sample).  The draw itself happens inside
    ``torch.multinomial`` (PyTorch 2.10, not part of /root/reference): for one sample per row on a device it takes
    ``argmax_j p_j / q_j`` with ``q ~ Exp(1)``; that rule is restated here and pinned on the GPU against
    ``torch.multinomial`` fed from the same generator state (tests/test_hip_kernels.py).  Returns the taken index, the
    one-hot action, log-prob [B, 1] and ``margin`` = best race / runner-up (a row with margin ~ 1 is a numerical tie)."""
    z = np.asarray(logits, np.float64)
    q = np.asarray(noise, np.float64)
    m = z.max(-1, keepdims=True)
    log_p = z - (m + np.log(np.exp(z - m).sum(-1, keepdims=True)))
    with np.errstate(divide="ignore"):
        race = np.exp(log_p) / q
    taken = race.argmax(-1)  # first maximum
    rows = np.arange(z.shape[0])
    ordered = np.sort(race, axis=-1)
    margin = ordered[:, -1] / ordered[:, -2] if z.shape[-1] > 1 else np.full(z.shape[0], np.inf)
    action = np.zeros(z.shape, np.float32)
    action[rows, taken] = 1.0
    return taken, action, log_p[rows, taken].astype(np.float32)[:, None], margin


def categorical_ppo_loss(advantage, old_logp, action, logits, ret, curr_value, old_value=None, *, clip=0.2,
                         value_clip=None, w_sur=1.0, w_val=0.5, w_ent=0.01, flip_clip_side=None):
    """The objective of :func:`ppo_loss` for a one-hot categorical policy, restated in numpy (float64 arithmetic, results
    rounded to float32): cusrl/nn/module/distribution.py:332-366 (``OneHotCategorical``: log-prob of the first arg-max
    of the one-hot action under log-softmax(logits), entropy ``-sum p log p``), hook/on_policy/ppo.py:10-18,82-84,
    hook/on_policy/value.py:85-89,121-137.  Returns losses[3], logp, entropy, ratio, d_logits, d_value."""
    adv = np.asarray(advantage, np.float64).reshape(-1)
    old_logp = np.asarray(old_logp, np.float64).reshape(-1)
    z = np.asarray(logits, np.float64)
    B, A = z.shape
    ret, cv = np.asarray(ret, np.float64), np.asarray(curr_value, np.float64)
    D = ret.shape[-1]
    taken = np.asarray(action).argmax(-1)  # first maximum, like value.max(-1)[1]
    log_p = z - (z.max(-1, keepdims=True) + np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)))
    prob = np.exp(log_p)
    logp = log_p[np.arange(B), taken]
    # torch.distributions.Categorical.entropy (which OneHotCategorical delegates to) clamps the normalised logits to
    # finfo(float32).min before multiplying by the probabilities: a masked action (logit -inf) contributes 0, not NaN
    log_p_clamped = np.maximum(log_p, np.float64(np.finfo(np.float32).min))
    entropy = -(prob * log_p_clamped).sum(-1)
    ratio = np.exp(logp - old_logp)
    lo, hi = np.float64(np.float32(1.0 - clip)), np.float64(np.float32(1.0 + clip))
    inside = (ratio >= lo) & (ratio <= hi)
    clipped = np.clip(ratio, lo, hi)
    if flip_clip_side is not None:  # see ppo_loss_f64: rows within fp32 noise of a clip bound, taken on the other side
        flip = np.asarray(flip_clip_side, bool).reshape(-1)
        nearest = np.where(np.abs(ratio - lo) < np.abs(ratio - hi), lo, hi)
        clipped = np.where(flip, np.where(inside, nearest, ratio), clipped)
        inside = inside ^ flip
    s1, s2 = adv * ratio, adv * clipped
    surrogate = -np.minimum(s1, s2).mean() * w_sur
    d_ratio = np.where(s1 < s2, adv, np.where(s1 > s2, np.where(inside, adv, 0.0), 0.5 * adv + np.where(inside, 0.5 * adv, 0.0)))
    dlp = (-w_sur / B) * d_ratio * ratio
    onehot = np.zeros_like(z)
    onehot[np.arange(B), taken] = 1.0
    d_logits = dlp[:, None] * (onehot - prob) - (-w_ent / B) * prob * (log_p_clamped + entropy[:, None])
    e1 = cv - ret
    if value_clip is None:
        value_loss, g = (e1 * e1).mean() * w_val, 2.0 * e1
    else:
        ov = np.asarray(old_value, np.float64)
        dv = cv - ov
        e2 = (ov + np.clip(dv, -value_clip, value_clip)) - ret
        l1, l2 = e1 * e1, e2 * e2
        value_loss = np.maximum(l1, l2).mean() * w_val
        g2 = np.where((dv >= -value_clip) & (dv <= value_clip), 2.0 * e2, 0.0)
        g = np.where(l1 > l2, 2.0 * e1, np.where(l1 < l2, g2, e1 + 0.5 * g2))
    d_value = (w_val / (B * D)) * g
    entropy_loss = -entropy.mean() * w_ent
    f = np.float32
    return dict(losses=np.array([value_loss, surrogate, entropy_loss], f), logp=logp.astype(f)[:, None],
                entropy=entropy.astype(f)[:, None], ratio=ratio.astype(f)[:, None], d_logits=d_logits.astype(f),
                d_value=d_value.astype(f), clip_margin=np.minimum(np.abs(ratio - lo), np.abs(ratio - hi)))
