"""``RandomSampler`` / ``TemporalRandomSampler`` / ``AutoRandomSampler`` (cusrl/sampler/random_sampler.py:18-138).

Pinned twice: the oracle's numpy restatement of the index arithmetic against batches recorded from the reference
(golden ``random_sampler.npz``, runs on CPU), and the HIP path (window-index kernel + multi-leaf gather, lazy batches)
against the same recordings bit for bit, plus the reference's own known answers (cusrl_test/sampler/test_random_sampler.py).
"""

import numpy as np
import pytest
import torch

import oracle

DEV = "cuda:0"
SAMPLERS = {
    "random": ("RandomSampler", dict(num_batches=2, batch_size=11)),
    "temporal_full": ("TemporalRandomSampler", dict(num_batches=2, batch_size=4)),
    "temporal_2": ("TemporalRandomSampler", dict(num_batches=3, batch_size=5, sequence_len=2)),
    "auto": ("AutoRandomSampler", dict(num_batches=2, batch_size=6, sequence_len=3)),
}


def _ring(g, p):
    """Physical ring contents after the recorded pushes (oracle.buffer_push), cursor and fullness."""
    capacity, parallelism, pushes, _ = (int(v) for v in g[p + "shape"])
    keys = [k[len(p + "push/"):] for k in g.files if k.startswith(p + "push/")]
    storage = {k: np.zeros((capacity,) + g[p + "push/" + k].shape[1:], g[p + "push/" + k].dtype) for k in keys}
    for step in range(pushes):
        for k in keys:
            oracle.buffer_push(g[p + "push/" + k][step], storage[k], step % capacity)
    return storage, pushes % capacity, pushes >= capacity


def test_oracle_restates_the_reference_samplers(golden):
    g = golden("random_sampler")
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        capacity, parallelism, pushes, has_memory = (int(v) for v in g[p + "shape"])
        storage, cursor, full = _ring(g, p)
        valid = capacity if full else cursor
        for tag, (cls, kwargs) in SAMPLERS.items():
            torch.manual_seed(1000 + i)
            temporal = cls == "TemporalRandomSampler" or (cls == "AutoRandomSampler" and has_memory)
            for j in range(kwargs["num_batches"]):
                q = f"{p}{tag}_{j}_"
                assert bool(g[q + "temporal"]) == temporal
                if temporal:
                    length = min(kwargs.get("sequence_len") or valid, valid)
                    env = torch.randint(parallelism, (kwargs["batch_size"],)).numpy()
                    start = torch.randint(valid - length + 1, (kwargs["batch_size"],)).numpy()
                    slots = oracle.window_slots(start, env, length, capacity, parallelism, cursor if full else None)
                    for k, leaf in storage.items():
                        rows = oracle.gather_rows(leaf, slots.reshape(-1)).reshape(slots.shape + leaf.shape[2:])
                        assert np.array_equal(rows, g[q + k]), (i, tag, j, k)
                else:
                    slots = torch.randint(valid * parallelism, (kwargs["batch_size"],)).numpy()
                    for k, leaf in storage.items():
                        assert np.array_equal(oracle.gather_rows(leaf, slots), g[q + k]), (i, tag, j, k)


@pytest.mark.gpu
def test_hip_samplers_replay_the_reference_batches(golden):
    import cusrl_amd as cusrl
    from cusrl_amd import _native
    from cusrl_amd.template.buffer import LazyBatch

    g = golden("random_sampler")
    windows_before = _native.launch_counts.get("cusrl_window_indices", 0)
    for i in range(int(g["num_cases"])):
        p = f"c{i}_"
        capacity, parallelism, pushes, has_memory = (int(v) for v in g[p + "shape"])
        buffer = cusrl.Buffer(capacity, parallelism, device=DEV)
        keys = [k[len(p + "push/"):] for k in g.files if k.startswith(p + "push/")]
        for step in range(pushes):
            buffer.push({k: torch.from_numpy(g[p + "push/" + k][step].copy()).to(DEV) for k in keys})
        for tag, (cls, kwargs) in SAMPLERS.items():
            for lazy in (True, False):
                sampler = getattr(cusrl, cls)(**kwargs, index_device="cpu", lazy=lazy)
                torch.manual_seed(1000 + i)
                count = 0
                for j, (metadata, batch) in enumerate(sampler(buffer)):
                    q = f"{p}{tag}_{j}_"
                    assert metadata["temporal"] == bool(g[q + "temporal"]) and metadata["batch_index"] == j
                    assert metadata["total_batches"] == kwargs["num_batches"] and isinstance(batch, LazyBatch) == lazy
                    for k in keys:
                        assert np.array_equal(batch[k].cpu().numpy(), g[q + k]), (i, tag, j, k, lazy)
                    count += 1
                assert count == kwargs["num_batches"]
    assert _native.launch_counts.get("cusrl_window_indices", 0) > windows_before


@pytest.mark.gpu
def test_reference_known_answers_on_the_device():
    """cusrl_test/sampler/test_random_sampler.py:7-92, on a GPU buffer with the device generator."""
    import cusrl_amd as cusrl

    def filled(parallelism, steps, with_memory=True):
        buffer = cusrl.Buffer(capacity=4, parallelism=parallelism, device=DEV)
        for step in range(steps):
            observation = torch.tensor([[float(step + 10 * n)] for n in range(parallelism)], device=DEV)
            buffer.push({"observation": observation, **({"actor_memory": observation + 100} if with_memory else {})})
        return buffer

    _, batch = next(iter(cusrl.RandomSampler(num_batches=1, batch_size=32)(filled(1, 3, False))))
    assert set(batch["observation"].squeeze(-1).tolist()) <= {0.0, 1.0, 2.0}                      # valid prefix only
    metadata, batch = next(iter(cusrl.TemporalRandomSampler(num_batches=1, batch_size=1)(filled(1, 3))))
    assert metadata["temporal"] is True
    assert batch["observation"].flatten().tolist() == [0.0, 1.0, 2.0]                             # unfilled tail trimmed
    assert batch["actor_memory"].flatten().tolist() == [100.0, 101.0, 102.0]
    _, batch = next(iter(cusrl.TemporalRandomSampler(num_batches=1, batch_size=1)(filled(1, 6))))
    assert batch["observation"].flatten().tolist() == [2.0, 3.0, 4.0, 5.0]                        # chronological window
    _, batch = next(iter(cusrl.TemporalRandomSampler(num_batches=1, batch_size=1, sequence_len=2)(filled(1, 3))))
    assert tuple(batch["observation"].flatten().tolist()) in {(0.0, 1.0), (1.0, 2.0)}
    _, batch = next(iter(cusrl.TemporalRandomSampler(num_batches=1, batch_size=2, sequence_len=2)(filled(2, 6))))
    valid = {(2.0, 3.0), (3.0, 4.0), (4.0, 5.0), (12.0, 13.0), (13.0, 14.0), (14.0, 15.0)}
    for sequence, memory in zip(batch["observation"].squeeze(-1).T.tolist(), batch["actor_memory"].squeeze(-1).T.tolist()):
        assert tuple(sequence) in valid and tuple(memory) == tuple(v + 100.0 for v in sequence)
    metadata, _ = next(iter(cusrl.AutoRandomSampler(num_batches=1, batch_size=2, sequence_len=2)(filled(2, 6))))
    assert metadata["temporal"] is True
    metadata, _ = next(iter(cusrl.AutoRandomSampler(num_batches=1, batch_size=2)(filled(2, 6, False))))
    assert metadata["temporal"] is False
    with pytest.raises(ValueError, match="'sequence_len' must be positive or None"):
        cusrl.TemporalRandomSampler(1, 1, sequence_len=0)
    with pytest.raises(RuntimeError, match="non-empty buffer"):
        next(iter(cusrl.TemporalRandomSampler(1, 1)(cusrl.Buffer(4, 1, device=DEV))))
