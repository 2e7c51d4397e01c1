"""Small helpers: global seeding (cusrl/utils/misc.py:144-181), first-present lookup (dict_utils.py:149-156)."""

from __future__ import annotations

import os
import random
import re
from collections.abc import Mapping
from typing import Any

import numpy as np
import torch

from cusrl_amd.utils.config import CONFIG

__all__ = ["MISSING", "camel_to_snake", "get_first", "host_form", "set_global_seed"]

MISSING = object()


def set_global_seed(seed: int | None, deterministic: bool = False) -> int:
    """Seed python / numpy / torch with ``seed + rank`` so every rank shards its own env stream."""
    if seed is None:
        seed = 42 if deterministic else int.from_bytes(os.urandom(4), "big")
    rank_seed = seed + CONFIG.rank
    random.seed(rank_seed)
    np.random.seed(rank_seed % (2**32))
    torch.manual_seed(rank_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(rank_seed)
    os.environ["PYTHONHASHSEED"] = str(rank_seed)
    if deterministic:
        torch.use_deterministic_algorithms(True)
    CONFIG.seed = seed
    return seed


def host_form(what: str) -> None:
    """Gate in front of the torch-op (host) form of a hook's device work.  The product has NO CPU path for the hot path: on
    an MI355X every hook takes its HIP entry point, and a CPU tensor reaching one of these places raises — unless the
    process opted in with ``CUSRL_HOST_FORMS=1``, which only the repository's own test infrastructure does (the host-logic
    tests and the gloo workers run in processes without a GPU and exercise the hooks' bookkeeping there)."""
    if os.environ.get("CUSRL_HOST_FORMS") != "1":
        raise RuntimeError(
            f"cusrl_amd: {what} received CPU tensors; the rollout + PPO-update hot path only runs as HIP kernels on an MI355X "
            "(no CPU fallback by design).  Test infrastructure without a GPU sets CUSRL_HOST_FORMS=1.")


def get_first(data: Mapping[str, Any], *keys: str) -> Any:
    """Value of the first key that is present and not None."""
    for key in keys:
        if (value := data.get(key)) is not None:
            return value
    raise KeyError(f"None of {keys} was found")


_CAMEL_1 = re.compile(r"(.)([A-Z][a-z]+)")
_CAMEL_2 = re.compile(r"([a-z0-9])([A-Z])")


def camel_to_snake(name: str) -> str:
    return _CAMEL_2.sub(r"\1_\2", _CAMEL_1.sub(r"\1_\2", name)).lower()
