"""Rollout buffer on HBM with one-launch append and one-launch minibatch gather.

Same public surface as ``cusrl.template.Buffer`` / ``Sampler`` (cusrl/template/buffer.py:16-207): circular
``[capacity, parallelism, ...]`` leaves keyed by the dotted path of a nested field, ``push`` / ``sample`` /
mapping access, identical validation messages.  What differs is how bytes move:

* ``push`` collects every leaf of the transition and issues ONE ``cusrl_buffer_push`` launch (the reference
  issues one index_put per leaf per step, buffer.py:146);
* ``gather`` (used by the minibatch samplers) issues ONE ``cusrl_gather_rows`` launch for all leaves (the
  reference runs one advanced-indexing kernel per leaf per minibatch, mini_batch_sampler.py:77,89).

Leaves are laid out exactly as the reference does — contiguous ``[T, N, C]`` with the env axis second — so the
env axis is unit-stride for C = 1 leaves (reward, value, flags) and rows are C*4 contiguous bytes otherwise.
"""

from __future__ import annotations

import os
from collections.abc import Callable, Iterator, Mapping, MutableMapping, Sequence
from typing import Any

import torch

from cusrl_amd import ops
from cusrl_amd.utils.config import device as resolve_device
from cusrl_amd.utils.nest import get_schema, iterate_nested, reconstruct_nested

__all__ = ["Buffer", "LazyBatch", "Sampler"]


def _native_max_fields() -> int:
    from cusrl_amd import _native

    return _native.MAX_FIELDS


class LazyBatch(dict):
    """A minibatch whose top-level fields are gathered on first access.

    The reference gathers EVERY stored leaf for every minibatch (mini_batch_sampler.py:77,89 — 1118 B per sample for
    the ``ppo`` buffer) although one PPO step reads about half of them (observation, action, old log-prob, advantage,
    return, value, done: 520 B).  A ``LazyBatch`` still *exposes* every field — ``batch["reward"]``, ``"x" in batch``,
    iteration, ``**batch`` all behave like the reference's dict — but a field's rows are only moved when somebody
    reads it.  Fields read by earlier batches of the same consumer (``hot``: a set the sampler / captured step owns) are
    fetched up front in ONE launch; anything else costs one extra launch on first access and joins ``hot``.  The
    very first batch of a consumer (empty ``hot``) gathers everything at once, exactly like the reference.

    Validity: pending fields read the buffer and the sampler's index slice *when accessed*; the sampler expires a batch
    when it moves on (its index storage is redrawn in place for the next epoch, mini_batch_sampler.py:67-68), after which
    reading a never-read field raises.  Fields already read stay valid forever, like the reference's fresh tensors.
    """

    __slots__ = ("_buffer", "_indices", "_temporal", "_pending", "_hot", "_expired", "_own", "_lead_shape")

    def __init__(self, buffer: "Buffer", indices: torch.Tensor, temporal: bool, hot: set | None,
                 lead_shape: tuple[int, ...] | None = None, preloaded: dict | None = None):
        super().__init__()
        self._buffer, self._indices, self._temporal = buffer, indices, temporal
        self._lead_shape = lead_shape
        self._hot = hot if hot is not None else set()
        self._expired = False
        self._own: set = set()  # keys the consumer wrote itself: reading them back says nothing about the buffer
        names = list(buffer.schema)
        if preloaded is not None:
            # fields somebody already gathered for exactly these indices (a captured step's prefetch, template/graphs.py):
            # they are simply there; every other field of the buffer stays available on first access
            dict.update(self, preloaded)
            self._pending = dict.fromkeys(name for name in names if name not in preloaded)
            return
        first = not self._hot
        eager = names if first else [name for name in names if name in self._hot]
        self._pending = dict.fromkeys(name for name in names if name not in eager)
        if eager:
            dict.update(self, self._gather(eager))

    # ---- materialisation
    def _gather(self, names):
        if self._lead_shape is None:  # (stand-in buffers of the host-logic tests only know the three-argument form)
            return self._buffer.gather(self._indices, self._temporal, fields=names)
        return self._buffer.gather(self._indices, self._temporal, fields=names, lead_shape=self._lead_shape)

    def _fetch(self, names):
        if self._expired:
            raise RuntimeError(
                f"batch field(s) {sorted(names)} were never read while this minibatch was current and its sampler has "
                "moved on; read them before advancing the sampler, or build the sampler with lazy=False")
        dict.update(self, self._gather(list(names)))
        for name in names:
            self._pending.pop(name, None)

    def _fetch_all(self):
        if self._pending:
            self._fetch(list(self._pending))

    def expire(self):
        """Called by the sampler when it advances: pending fields can no longer be produced."""
        self._expired = True
        self._buffer = self._indices = None

    # ---- dict protocol
    def __missing__(self, key):
        if key in self._pending:
            self._hot.add(key)
            self._fetch([key])
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __getitem__(self, key):
        try:
            value = dict.__getitem__(self, key)
        except KeyError:
            return self.__missing__(key)
        if key not in self._own:
            self._hot.add(key)
        return value

    def get(self, key, default=None):
        if dict.__contains__(self, key) or key in self._pending:
            return self[key]
        return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._pending

    def __setitem__(self, key, value):
        self._pending.pop(key, None)  # an overwritten field never needs its rows
        self._own.add(key)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        if key in self._pending:
            del self._pending[key]
            return
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        if key in self._pending:
            self._fetch([key])
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default

    def update(self, *args, **kwargs):
        # through __setitem__: a field the consumer overwrites must leave `_pending`, or a later whole-batch access
        # (iteration, **batch, copy) would fetch the buffer's rows over the consumer's value
        for key, value in dict(*args, **kwargs).items():
            self[key] = value

    def __ior__(self, other):
        self.update(other)
        return self

    def __or__(self, other):
        merged = self.copy()
        merged.update(other)
        return merged

    def __ror__(self, other):
        merged = dict(other)
        merged.update(self.copy())
        return merged

    def __iter__(self):
        self._fetch_all()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + sum(1 for key in self._pending if not dict.__contains__(self, key))

    def keys(self):
        self._fetch_all()
        return dict.keys(self)

    def values(self):
        self._fetch_all()
        return dict.values(self)

    def items(self):
        self._fetch_all()
        return dict.items(self)

    def copy(self):
        self._fetch_all()
        return dict(self)

    def __eq__(self, other):
        self._fetch_all()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        return f"LazyBatch({dict.__repr__(self)}, pending={list(self._pending)})"


class Buffer(MutableMapping):
    def __init__(self, capacity: int, parallelism: int, device: str | torch.device | None = None):
        self.capacity: int = capacity
        self.parallelism: int = parallelism
        self.device = resolve_device(device)
        self.cursor = 0
        self.full = False
        self.schema: dict[str, Any] = {}
        self.storage: dict[str, torch.Tensor] = {}
        # by-products of a kernel that remain valid until someone else touches the field (e.g. the advantage
        # {sum, sumsq} partials the GAE kernel emits for the normalisation hook)
        self._derived: dict[str, Any] = {}
        self._push_plan = None
        self.pending_epilogue = None  # ops.PendingStepEpilogue of the env step being appended (see push)
        # Leaves interleaved into one record per slot for the minibatch gather (ops.RecordPack), kept coherent LEAF BY LEAF:
        # `_record_clean[leaf]` is the leaf tensor's version counter at the moment the record mirrored it.  Any in-place
        # edit through torch — by whoever holds an alias handed out by `buffer[key]`, like the reference allows
        # (buffer.py:119-122) — bumps that counter; this package's raw kernels either go through `field()` / `push()`
        # (which drop the entry) or bump the counter themselves (ops._modified_in_place).  `prepare_sampling()` re-packs
        # exactly the stale leaves, `gather()` reads a leaf through the record only while it is provably current.
        self.pack_narrow_leaves = True
        self.pack_hot_fields = True
        # the record pays when sampled rows miss the caches (a random row costs whole 128-byte lines out of HBM); while the
        # packed leaves fit L2 + Infinity Cache the plain per-leaf gather is faster and needs no pack launch at all
        # (config 2: 25 MB, 5.5 vs 6.7 us per minibatch; measured break-even between 64 and 256 MB)
        self.record_threshold_bytes = int(os.environ.get("CUSRL_RECORD_THRESHOLD_BYTES", 128 << 20))
        self._pack = None
        self._pack_hot = None
        self._hot_fields: set[str] = set()
        self._hot_dirty = False
        self._record_clean: dict[str, int] = {}
        self._through = None  # (pack, storage layout) -> write-through arguments of the steady-state push
        # bumped whenever a storage tensor or the packed record is (re)allocated: captured hipGraphs bake those
        # addresses in and compare this number before replaying (template/graphs.py)
        self.layout_version = 0
        self._storage_epoch = 0
        self._pack_key = None

    # ------------------------------------------------------------------ bookkeeping
    def get_parallelism(self) -> int:
        return self.parallelism

    def clear(self):
        self.cursor = 0
        self.full = False
        self.storage.clear()
        self.schema.clear()
        self._derived.clear()
        self._push_plan = None
        self._pack = None
        self._record_clean.clear()
        self._through = None
        self._hot_dirty = bool(self._hot_fields)  # the field names stay known; their leaves are re-resolved
        self._storage_changed()

    def reset_cursor(self):
        self.cursor = 0

    def resize(self, capacity: int):
        if capacity != self.capacity:
            self.clear()
            self.capacity = capacity

    # ------------------------------------------------------------------ mapping protocol (top-level field names)
    def __iter__(self):
        yield from self.schema

    def __len__(self):
        return len(self.schema)

    def __contains__(self, key):
        return key in self.schema

    def __getitem__(self, key):
        # hands out the storage tensors themselves: in-place edits by hooks are visible (advantage.py:102) — also to the
        # per-slot record, through the tensors' version counters (no matter when the edit happens)
        self._derived.pop(key, None)
        return reconstruct_nested(self.storage, self.schema[key])

    def get(self, key, default=None):
        if (schema := self.schema.get(key)) is None:
            return default
        self._derived.pop(key, None)
        return reconstruct_nested(self.storage, schema)

    def _touch(self, name: str):
        """A writer that torch's version counters cannot see (a raw kernel) is about to write field ``name``."""
        schema = self.schema.get(name)
        if schema is not None and self._record_clean:
            for _, key in iterate_nested(schema):
                self._record_clean.pop(key, None)

    def __setitem__(self, name, data):
        """Register or overwrite a whole field; every leaf must be ``[capacity, parallelism, ...]``."""
        if data is None:
            return
        self._check_schema(name, data)
        self._derived.pop(name, None)
        self._touch(name)
        for key, value in iterate_nested(data, name):
            value = self._as_tensor(value)
            self._validate_field_shape(key, value.shape)
            storage = self.storage.get(key)
            if storage is None:
                storage = self.storage[key] = torch.zeros_like(value, device=self.device)
                self._storage_changed()
            if storage.data_ptr() != value.data_ptr():
                storage.copy_(value)

    def __delitem__(self, name: str):
        if name not in self.schema:
            raise KeyError(f"Field '{name}' was not found")
        for _, key in iterate_nested(self.schema[name]):
            del self.storage[key]
        del self.schema[name]
        self._derived.pop(name, None)
        self._pack = None
        self._record_clean.clear()
        self._through = None
        self._storage_changed()

    # ------------------------------------------------------------------ extensions used by the HIP hooks
    def field(self, name: str, like: torch.Tensor) -> torch.Tensor:
        """Storage tensor of the single-leaf field ``name``, allocated (uninitialised) like ``like`` on first use,
        so kernels write their results straight into buffer-owned memory (no ``copy_`` as in buffer.py:101)."""
        storage = self.storage.get(name)
        if storage is None:
            self._validate_field_shape(name, like.shape)
            storage = self.storage[name] = torch.empty_like(like, device=self.device)
            self.schema[name] = name
            self._storage_changed()
        self._derived.pop(name, None)
        self._record_clean.pop(name, None)
        return storage

    def set_derived(self, name: str, value: Any):
        self._derived[name] = value

    def take_derived(self, name: str) -> Any:
        return self._derived.pop(name, None)

    # ------------------------------------------------------------------ a1: append one step
    def push(self, data: Mapping[str, Any]):
        """Append one time step of every field; each leaf is ``[parallelism, ...]``.

        The first write of a leaf fixes its nested schema and allocates ``[capacity, parallelism, ...]`` with the
        pushed dtype; ``None`` fields are skipped; after ``capacity`` pushes the buffer is ``full`` and the
        cursor wraps to 0 (buffer.py:124-151).
        """
        # a step epilogue the trainer handed over (ops.PendingStepEpilogue): issued WITH the steady-state append as one
        # launch, or on its own in front of any other path — the append copies the `done` flag it produces
        pending, self.pending_epilogue = self.pending_epilogue, None
        if self._fast_push(data, pending):
            return
        if pending is not None:
            pending.launch()
        pairs = []
        for name, nested_value in data.items():
            if nested_value is None:
                continue
            self._check_schema(name, nested_value)
            self._derived.pop(name, None)
            self._touch(name)
            for key, value in iterate_nested(nested_value, name):
                value = self._as_tensor(value)
                storage = self.storage.get(key)
                if storage is None:
                    self._validate_step_shape(key, value.shape)
                    storage = self.storage[key] = value.new_zeros(self.capacity, *value.shape)
                    self._storage_changed()
                elif value.shape != storage.shape[1:]:
                    raise ValueError(
                        f"Shape mismatch for field '{key}': expected {tuple(storage.shape[1:])}, got {tuple(value.shape)}"
                    )
                if value.dtype != storage.dtype:
                    value = value.to(storage.dtype)
                if not value.is_contiguous():
                    value = value.contiguous()
                pairs.append((value, storage))
        if pairs:
            ops.require_device(pairs[0][1], "buffer storage")
            ops.buffer_push(pairs, self.cursor, self.parallelism)
        self._advance()
        self._build_push_plan(data)

    def _advance(self):
        self.cursor += 1
        if self.cursor == self.capacity:
            self.full = True
            self.cursor = 0

    # ---- steady-state append: the transition has the same fields every step, so everything that does not change
    # (schema, shapes, destination pointers, row sizes) is resolved once and a push is "fill 11 source pointers + launch"
    def _build_push_plan(self, data: Mapping[str, Any]):
        self._push_plan = None
        if len(self.storage) > _native_max_fields():
            return
        names, absent, nested_sizes, specs, storages, keys = [], [], [], [], [], []
        for name, nested_value in data.items():
            names.append(name)
            if nested_value is None:
                absent.append(name)
                continue
            if isinstance(nested_value, (Mapping, tuple, list)):
                nested_sizes.append((name, len(nested_value)))
            for key, value in iterate_nested(nested_value, name):
                if not isinstance(value, torch.Tensor):
                    return  # numpy / python inputs take the converting path
                path = tuple(int(part) if part.isdigit() else part for part in key.split(".")[1:])
                specs.append((name, path, value.shape, value.dtype))
                storages.append((self.storage[key], value.shape))
                keys.append(key)
        table = ops.make_push_table(storages)
        fields = [table[i] for i in range(len(specs))]  # ctypes views of the array slots: `.src` writes go straight in
        leaves = tuple((name, path, shape, dtype, field) for (name, path, shape, dtype), field in zip(specs, fields))
        self._push_plan = (tuple(names), tuple(absent), tuple(nested_sizes), leaves, table, tuple(self.storage.items()),
                           tuple(keys))
        self._through = None

    def _fast_push(self, data: Mapping[str, Any], pending=None) -> bool:
        plan = getattr(self, "_push_plan", None)
        if plan is None:
            return False
        names, absent, nested_sizes, leaves, table, storage_refs, keys = plan
        if tuple(data) != names:
            return False
        for name in absent:
            if data[name] is not None:
                return False
        device, tensor_type = self.device, torch.Tensor
        try:
            for name, path, shape, dtype, field in leaves:
                value = data[name]
                for part in path:
                    value = value[part]
                # a None / numpy / wrong-shape value fails here and takes the converting path
                if (type(value) is not tensor_type or value.shape != shape or value.dtype != dtype
                        or value.device != device or not value.is_contiguous()):
                    return False
                field.src = value.data_ptr()
        except (KeyError, IndexError, TypeError, ValueError):
            return False
        for name, size in nested_sizes:  # nested containers may have gained leaves the plan does not know
            if len(data[name]) != size:
                return False
        storage = self.storage
        if len(storage_refs) != len(storage) or any(storage.get(k) is not t for k, t in storage_refs):
            return False  # a field was replaced / added behind our back
        if self._derived:
            for name in names:
                self._derived.pop(name, None)
        through = self._push_through(keys)
        if pending is not None:
            done_field = next((i for i, (name, path, *_rest) in enumerate(leaves) if name == "done" and not path), -1)
            if through is None and done_field >= 0 and data["done"] is pending.done_out:
                pending.launch_with_push(table, len(leaves), done_field, self.cursor, self.parallelism)
                self._advance()
                return True
            pending.launch()
        ops.push_table(table, len(leaves), self.cursor, self.parallelism, through)
        self._advance()
        return True

    def _push_through(self, keys):
        """Write-through arguments of the steady-state push and the record bookkeeping of one push: the wide leaves of
        the current record are stored into it by the push kernel itself (they stay current — the pack at update time then
        only moves the narrow leaves); every other pushed leaf stops being mirrored."""
        pack, clean = self._pack, self._record_clean
        if pack is None:
            return None
        cached = self._through
        if cached is None or cached[0] is not pack or cached[1] is not keys:
            offsets = pack.through_offsets(keys)
            kept = frozenset(key for key, offset in zip(keys, offsets) if offset >= 0) if offsets is not None else frozenset()
            cached = self._through = (pack, keys, None if offsets is None else (pack.record, pack.record_bytes, offsets), kept)
        if clean:
            kept = cached[3]
            for key in keys:
                if key not in kept:
                    clean.pop(key, None)
        return cached[2]

    def replay_push(self):
        """Host bookkeeping of a steady-state :meth:`push` whose launch was replayed from a hipGraph (the cursor the
        graph was captured at is baked into it; the caller keys its graphs on ``cursor``)."""
        plan = self._push_plan
        if plan is None:
            raise RuntimeError("replay_push: no steady-state append has been planned for this buffer")
        if self._derived:
            for name in plan[0]:
                self._derived.pop(name, None)
        self._push_through(plan[6])  # the same record bookkeeping as the push that was captured
        self._advance()

    # ------------------------------------------------------------------ a7/a8: sampling
    def sample(self, sampler: Callable[[str, torch.Tensor], torch.Tensor]) -> dict[str, Any]:
        """Generic per-leaf callback form of the reference (buffer.py:153-162)."""
        batch = {key: sampler(key, tensor) for key, tensor in self.storage.items()}
        return reconstruct_nested(batch, self.schema)

    def prepare_sampling(self, hot_fields=None) -> bool:
        """Refresh the per-slot record (``cusrl_pack_rows``) if anything could have written to its leaves since it was
        built; True when that launched something (on the current stream).  The samplers call this once per pass before the
        first minibatch; it is a flag check when nothing changed.
        ``hot_fields``: the top-level fields the consumer is known to read (a sampler's ``hot_fields``); once known, the
        record holds exactly their leaves — wide ones included — so a sampled slot is two memory lines; before that
        (first pass) it holds the narrow leaves.  Must run OUTSIDE hipGraph capture (captured steps only *read* it)."""
        if hot_fields and self.pack_hot_fields:
            # the union over all consumers, so that two samplers with different appetites do not re-plan the record in turns
            known, schema = self._hot_fields, self.schema
            fresh = [name for name in hot_fields if name not in known and name in schema]
            if fresh:
                known.update(fresh)
                self._hot_dirty = True
        if self._hot_dirty:
            self._hot_dirty = False
            hot = None
            if self._hot_fields and self.pack_hot_fields:
                hot = tuple(key for name in self.schema if name in self._hot_fields for _, key in iterate_nested(self.schema[name]))
            if hot != self._pack_hot:
                self._pack_hot = hot
                self._record_clean.clear()
        if not self.pack_narrow_leaves or self.device.type != "cuda":
            self._pack = None
            return False
        pack = self._pack
        key = self._pack_plan_key()
        if key != self._pack_key:  # leaves were (re)allocated or the plan changed: plan again
            self._pack_key = key
            names = ops.RecordPack.plan(self.storage, self._pack_hot)
            packed_bytes = sum(ops._row_bytes(self.storage[name], 2) for name in names) * self.capacity * self.parallelism
            if not names or packed_bytes < self.record_threshold_bytes:
                if pack is not None:
                    self.layout_version += 1
                self._pack = None
                self._record_clean.clear()
                return False
            wanted = tuple((name, self.storage[name].data_ptr(), ops._row_bytes(self.storage[name], 2)) for name in names)
            if pack is None or set(pack.key) != set(wanted):
                pack = self._pack = ops.RecordPack({name: self.storage[name] for name in names})
                self._record_clean.clear()
                self._through = None
                self.layout_version += 1
        if pack is None:
            return False
        clean, storage = self._record_clean, self.storage
        stale = [name for name in pack.leaves if clean.get(name) != storage[name]._version]
        if stale:
            pack.build(None if len(stale) == len(pack.leaves) else stale)
            for name in stale:
                clean[name] = storage[name]._version
        return bool(stale)

    def _pack_plan_key(self):
        return (self._pack_hot, self.record_threshold_bytes, self._storage_epoch)

    def _storage_changed(self):
        """A storage tensor was allocated or dropped: captured graphs bake addresses (layout_version), the record plan
        depends on the set of leaves (_storage_epoch)."""
        self.layout_version += 1
        self._storage_epoch += 1

    def _mirrored(self, pack, key: str) -> bool:
        """Does the record hold leaf ``key`` as it is right now?"""
        return key in pack.offsets and self._record_clean.get(key) == self.storage[key]._version

    def gather(self, indices: torch.Tensor, temporal: bool = False, fields: Sequence[str] | None = None,
               lead_shape: tuple[int, ...] | None = None, out: dict[str, torch.Tensor] | None = None) -> dict[str, Any]:
        """``flatten(0, 1)[indices]`` (or ``[:, indices]`` when ``temporal``) of every leaf — or of the leaves of the
        top-level ``fields`` only — in one launch; narrow leaves come through the packed record while it is current.
        ``lead_shape``: view the gathered rows as ``[*lead_shape, ...]`` (the ``[sequence_len, batch]`` windows of the
        temporal random sampler, whose flat slot list is ``sequence_len * batch`` long).  ``out``: destinations by leaf key the
        caller keeps from call to call (a captured step's persistent batch tensors): a leaf found there is gathered into that
        tensor, a leaf missing there gets a fresh tensor which is left in ``out``."""
        if fields is None:
            keys, schema = list(self.storage), self.schema
        else:
            schema = {name: self.schema[name] for name in fields}
            keys = [key for name in fields for _, key in iterate_nested(self.schema[name])]
        if out is not None:
            batch = indices.numel()
            lead = (self.capacity, batch) if temporal else (batch,)
            for key in keys:
                if key not in out:
                    leaf = self.storage[key]
                    out[key] = torch.empty(lead + tuple(leaf.shape[2:]), dtype=leaf.dtype, device=leaf.device)
        pack = self._pack
        packed = [key for key in keys if pack is not None and self._mirrored(pack, key)]
        plain = [key for key in keys if key not in packed] if packed else keys
        if packed and len(plain) <= _native_max_fields():
            outputs, packed_outputs = ops.gather_rows_packed(
                [self.storage[k] for k in plain], pack, packed, indices, self.capacity, self.parallelism, temporal,
                out=None if out is None else [out[k] for k in plain], packed_out=None if out is None else [out[k] for k in packed])
            gathered = dict(zip(plain, outputs))
            gathered.update(zip(packed, packed_outputs))
        else:
            outputs = ops.gather_rows([self.storage[k] for k in keys], indices, self.capacity, self.parallelism, temporal,
                                      out=None if out is None else [out[k] for k in keys])
            gathered = dict(zip(keys, outputs))
        if lead_shape is not None:
            gathered = {key: rows.view(tuple(lead_shape) + tuple(rows.shape[1:])) for key, rows in gathered.items()}
        return reconstruct_nested(gathered, schema)

    def gather_lazy(self, indices: torch.Tensor, temporal: bool = False, hot: set | None = None,
                    lead_shape: tuple[int, ...] | None = None, preloaded: dict | None = None) -> LazyBatch:
        """The minibatch as a :class:`LazyBatch`: the ``hot`` fields now (one launch) — or the ``preloaded`` ones as they are,
        no launch — the rest on first access."""
        return LazyBatch(self, indices, temporal, hot, lead_shape, preloaded)

    # ------------------------------------------------------------------ validation (messages as in the reference)
    def _as_tensor(self, data) -> torch.Tensor:
        return torch.as_tensor(data, device=self.device)

    def _validate_step_shape(self, name: str, shape: Sequence[int]):
        if len(shape) < 2:
            raise ValueError(f"A step of field '{name}' must have shape [parallelism, ...]")
        if shape[0] != self.parallelism:
            raise ValueError(f"Parallelism mismatch for field '{name}': expected {self.parallelism}, got {shape[0]}")

    def _validate_field_shape(self, name: str, shape: Sequence[int]):
        if len(shape) < 3:
            raise ValueError(f"Field '{name}' must have shape [capacity, parallelism, ...]")
        if shape[0] != self.capacity:
            raise ValueError(f"Capacity mismatch for field '{name}': expected {self.capacity}, got {shape[0]}")
        if shape[1] != self.parallelism:
            raise ValueError(f"Parallelism mismatch for field '{name}': expected {self.parallelism}, got {shape[1]}")

    def _check_schema(self, name: str, data):
        current = get_schema(data, name)
        known = self.schema.get(name)
        if known is None:
            self.schema[name] = current
        elif known != current:
            raise ValueError(f"Schema mismatch for field '{name}': expected '{known}', got '{current}'")


class Sampler:
    """Base sampler: one batch = the whole buffer, no copy (buffer.py:193-207)."""

    def __call__(self, buffer: Buffer) -> Iterator[tuple[dict[str, Any], dict[str, Any]]]:
        yield {}, buffer.sample(lambda _name, tensor: tensor)
