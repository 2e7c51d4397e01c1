"""Measured GEMM kernel selection for the actor-critic's library GEMMs (PyTorch TunableOp).

The MLP / recurrent GEMMs stay rocBLAS / hipBLASLt calls (BASELINE.json north_star).  For the odd shapes of this workload
— ``[24576, 48] x [48, 256]``, batched ``[16, 256, 1536] x [16, 1536, 128]`` weight-gradient slabs, 12- and 1-column
heads — the libraries' default heuristic is not always the fastest kernel they ship; ``scripts/tune_gemms.py`` times
the candidates on the MI355X once and records the winners in ``cusrl_amd/tuned_gemms_gfx950.csv``.  Loading that file
only changes WHICH library kernel runs a shape (TunableOp rejects the file when the installed ROCm / hipBLASLt /
rocBLAS / PyTorch versions differ from the ones it was recorded with; shapes not in it keep the default)."""

from __future__ import annotations

import os
from pathlib import Path

import torch

__all__ = ["enable_tuned_gemms", "TUNED_GEMMS_FILE"]

TUNED_GEMMS_FILE = Path(__file__).resolve().parent.parent / "tuned_gemms_gfx950.csv"
_state: dict[str, bool | None] = {"enabled": None}


def enable_tuned_gemms(path: str | os.PathLike | None = None) -> bool:
    """Idempotent; returns whether a selection file is active.  ``CUSRL_TUNED_GEMMS=0`` leaves torch untouched,
    ``CUSRL_TUNED_GEMMS=<file>`` loads another selection (A/B runs of a re-tuned file)."""
    if _state["enabled"] is not None and path is None:
        return bool(_state["enabled"])
    active = False
    choice = os.environ.get("CUSRL_TUNED_GEMMS", "1")
    file = Path(path) if path is not None else (TUNED_GEMMS_FILE if choice in ("0", "1") else Path(choice))
    if choice != "0" and torch.cuda.is_available() and file.exists():
        import torch.cuda.tunable as tunable

        if not tunable.tuning_is_enabled() or not tunable.is_enabled():  # a user's own TunableOp session is left alone
            tunable.enable(True)
            tunable.tuning_enable(False)
            active = bool(tunable.read_file(str(file)))
            if not active:
                tunable.enable(False)
    _state["enabled"] = active
    return active
