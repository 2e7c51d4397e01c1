"""ctypes binding of ``libcusrl_hip.so`` (C ABI declared in ``include/cusrl_hip.h``).

The library is the product: there is no CPU or eager fallback.  If it has not been built
(``python __graft_entry__.py`` / ``__graft_entry__.build()``) every hot-path call raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_void_p
from pathlib import Path

# (CUSRL_HIP_LIBRARY: another build of the same library — A/B runs of compile-time variants, profiles/r05/loss_variants_ab.txt)
LIB_PATH = Path(os.environ.get("CUSRL_HIP_LIBRARY") or Path(__file__).resolve().parent / "libcusrl_hip.so")
ABI_VERSION = 6
MAX_FIELDS = 24
MAX_PACKED = 16


class Field(Structure):
    """``cusrl_field_t`` — one buffer leaf of a multi-leaf launch."""

    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("row_bytes", c_int64)]


class PackedField(Structure):
    """``cusrl_packed_field_t`` — one narrow leaf's place inside the per-slot record."""

    _fields_ = [("ptr", c_void_p), ("offset", ctypes.c_int32), ("width", ctypes.c_int32)]


class GradPiece(Structure):
    """``cusrl_grad_piece_t`` — one parameter's slot of the flat gradient buffer and what to sum into it."""

    _fields_ = [("src", c_void_p), ("offset", c_int64), ("numel", c_int64), ("splits", c_int64), ("row_stride", c_int64)]


class NativeError(RuntimeError):
    pass


_lib = None

_P = c_void_p
_SIGNATURES = {
    "cusrl_abi_version": (c_int, []),
    "cusrl_error_string": (c_char_p, [c_int]),
    "cusrl_set_option": (c_int, [c_char_p, c_int64]),
    "cusrl_get_option": (c_int, [c_char_p, POINTER(c_int64)]),
    "cusrl_buffer_push": (c_int, [POINTER(Field), c_int, c_int64, c_int64, _P]),
    "cusrl_buffer_push_through": (c_int, [POINTER(Field), c_int, c_int64, c_int64, _P, c_int64, POINTER(ctypes.c_int32), _P]),
    "cusrl_next_value": (c_int, [_P, _P, _P, _P, c_float, c_int, _P, _P, c_int64, c_int64, c_int64, _P]),
    "cusrl_flag_blocks": (c_int64, [c_int64]),
    "cusrl_compact_flags": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P]),
    "cusrl_scatter_rows": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P]),
    "cusrl_splice_rows": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, _P]),
    "cusrl_gae": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_double, c_double, c_double, _P]),
    "cusrl_gae_num_partials": (c_int64, [c_int64, c_int64, c_int64]),
    "cusrl_col_stats": (c_int, [_P, c_int64, c_int64, _P, _P]),
    "cusrl_col_stats_num_partials": (c_int64, [c_int64, c_int64]),
    "cusrl_stats_finalize": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "cusrl_normalize": (c_int, [_P, _P, _P, c_float, c_int64, c_int64, _P]),
    "cusrl_normalize_from_partials": (c_int, [_P, _P, c_int64, c_int64, c_float, c_int64, c_int64, _P, _P, _P]),
    "cusrl_merge_mean_var": (c_int, [_P, c_int64, c_int64, _P, _P, _P]),
    "cusrl_normalize_from_gathered": (c_int, [_P, _P, c_int64, c_float, c_int64, c_int64, _P, _P, _P]),
    "cusrl_gather_rows": (c_int, [POINTER(Field), c_int, _P, c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_pack_rows": (c_int, [POINTER(PackedField), c_int, _P, c_int64, c_int64, _P]),
    "cusrl_pack_rows_owned": (c_int, [POINTER(PackedField), c_int, _P, c_int64, c_int64, c_int, c_int, _P]),
    "cusrl_gather_rows_packed": (c_int, [POINTER(Field), c_int, _P, c_int64, POINTER(PackedField), c_int, _P, c_int64, c_int64,
                                         c_int64, c_int, _P]),
    "cusrl_window_indices": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int64, _P]),
    "cusrl_ppo_loss_fwd_bwd": (
        c_int,
        [_P] * 8 + [c_int64] * 3 + [c_double] * 5 + [_P] * 8 + [_P, c_int64, _P, c_int, _P],
    ),
    "cusrl_ppo_loss_categorical_fwd_bwd": (c_int, [_P] * 7 + [c_int64] * 3 + [c_double] * 5 + [_P] * 7 + [_P, c_int, _P]),
    "cusrl_ppo_loss_num_partials": (c_int64, [c_int64]),
    "cusrl_ppo_loss_std_partial_rows": (c_int64, [c_int64]),
    "cusrl_ppo_loss_blocks": (c_int64, [c_int64, c_int64]),
    "cusrl_value_loss_fwd_bwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_double, c_double, _P, _P, _P, c_int, _P]),
    "cusrl_value_loss_blocks": (c_int64, [c_int64, c_int64]),
    "cusrl_normal_sample_logp": (c_int, [_P] * 5 + [c_int64, c_int64, c_int64, _P, _P, _P, _P]),
    "cusrl_categorical_sample_logp": (c_int, [_P] * 4 + [c_int64, c_int64, _P]),
    "cusrl_gru_gates_fwd": (c_int, [_P] * 6 + [c_int64, c_int64, c_int64, _P]),
    "cusrl_gru_gates_bwd": (c_int, [_P] * 7 + [c_int64, c_int64, c_int64, _P]),
    "cusrl_gru_gates_bwd_bias": (c_int, [_P] * 7 + [c_int64, c_int64, c_int64, _P, _P]),
    "cusrl_gru_bias_partial_rows": (c_int64, [c_int64]),
    "cusrl_gru_bias_supported": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "cusrl_lstm_gates_fwd": (c_int, [_P] * 8 + [c_int64, c_int64, c_int64, _P]),
    "cusrl_lstm_gates_bwd": (c_int, [_P] * 7 + [c_int64, c_int64, c_int64, _P]),
    "cusrl_rnn_cell_fwd": (c_int, [_P] * 6 + [c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_rnn_cell_bwd": (c_int, [_P] * 5 + [c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_episode_stats": (c_int, [_P] * 8 + [c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_step_epilogue": (c_int, [_P] * 12 + [c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_step_epilogue_max_envs": (c_int64, []),
    "cusrl_step_epilogue_push": (c_int, [_P] * 12 + [c_int64, c_int64, c_int64, c_int, POINTER(Field), c_int, c_int, c_int64, _P]),
    "cusrl_policy_stats": (c_int, [_P] * 7 + [c_int64, c_int64, c_int64, _P, _P, _P]),
    "cusrl_policy_stats_num_partials": (c_int64, [c_int64]),
    "cusrl_categorical_policy_stats": (c_int, [_P] * 5 + [c_int64, c_int64, c_int64, _P, _P, _P]),
    "cusrl_policy_terms_fwd": (c_int, [_P, _P, c_int64, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "cusrl_policy_terms_bwd": (c_int, [_P, _P, c_int64] + [_P] * 6 + [c_int64, c_int64, _P, _P, _P, _P]),
    "cusrl_policy_terms_std_partial_rows": (c_int64, [c_int64]),
    "cusrl_categorical_terms_fwd": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "cusrl_categorical_terms_bwd": (c_int, [_P] * 7 + [c_int64, c_int64, _P, _P]),
    "cusrl_relu_bwd_colsum": (c_int, [_P] * 5 + [c_int64, c_int64, _P]),
    "cusrl_colsum_num_partials": (c_int64, [c_int64, c_int64]),
    "cusrl_input_layer_bwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "cusrl_input_layer_supported": (c_int, [c_int64, c_int64]),
    "cusrl_input_layer_row_blocks": (c_int64, [c_int64, c_int64]),
    "cusrl_narrow_linear_bwd": (c_int, [_P] * 6 + [c_int64, c_int64, c_int64, c_int, _P]),
    "cusrl_narrow_linear_num_partials": (c_int64, [c_int64]),
    "cusrl_narrow_linear_supported": (c_int, [c_int64, c_int64]),
    "cusrl_narrow_linear_fwd": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_int64, _P]),
    "cusrl_mlp2_forward": (c_int, [_P, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "cusrl_mlp2_forward_supported": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "cusrl_clip_grad_norm": (c_int, [_P, c_int64, c_float, _P, _P, _P]),
    "cusrl_clip_grad_norm_num_partials": (c_int64, [c_int64]),
    "cusrl_assemble_gradients": (c_int, [POINTER(GradPiece), c_int64, _P, _P, _P]),
    "cusrl_assemble_gradients_blocks": (c_int64, [POINTER(GradPiece), c_int64]),
    "cusrl_grad_sumsq": (c_int, [_P, c_int64, _P, _P]),
    "cusrl_adam_step": (c_int, [_P] * 6 + [c_int64, c_double, c_double, c_double, c_double, c_int, c_int, _P, c_int64, c_float, _P, _P, _P, _P]),
    "cusrl_adam_step_window": (c_int, [_P] * 6 + [c_int64, c_double, c_double, c_double, c_double, c_int, c_int, _P, c_int64, _P, c_int64,
                                       c_float, _P, _P, _P, _P, _P]),
    "cusrl_adam_step_normed": (c_int, [_P] * 6 + [c_int64, c_double, c_double, c_double, c_double, c_int, c_int, _P, c_int64, _P,
                                       c_float, _P, _P, _P, _P, _P]),
    "cusrl_adam_step_normed_workspace_bytes": (c_int64, []),
    "cusrl_masked_col_stats": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "cusrl_masked_stats_num_partials": (c_int64, [c_int64, c_int64]),
    "cusrl_rms_merge": (c_int, [_P] * 7 + [c_float, c_double, c_int64, _P]),
    "cusrl_rms_normalize": (c_int, [_P, _P, _P, c_float, _P, c_int64, c_int64, _P]),
    "cusrl_rnd_reward": (c_int, [_P, _P, _P, _P, c_float, c_int64, c_int64, _P]),
    "cusrl_amp_style_reward": (c_int, [_P, _P, _P, c_float, c_int64, _P]),
    "cusrl_amp_style_reward_mean": (c_int, [_P, _P, _P, c_float, c_int64, _P, _P]),
    "cusrl_amp_prepare": (c_int, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, c_float, c_double,
                                  c_float, _P, _P, _P, _P]),
    "cusrl_amp_prepare_max_elements": (c_int64, []),
    "cusrl_amp_prepare_workspace": (c_int64, [c_int64, c_int64]),
    "cusrl_synthetic_env_step": (c_int, [ctypes.c_uint64, _P, c_int64, c_int64, c_int64, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "cusrl_accumulate_scalars": (c_int, [POINTER(c_void_p), c_int, _P, _P]),
    "cusrl_reward_shaping": (c_int, [_P, c_float, c_float, c_float, c_float, c_int, c_int, c_int64, _P]),
    "cusrl_mse_loss_fwd_bwd": (c_int, [_P, _P, c_int64, _P, _P, _P, _P]),
    "cusrl_mse_loss_num_partials": (c_int64, [c_int64]),
    "cusrl_sumsq_fwd_bwd": (c_int, [_P, c_int64, c_double, c_double, _P, _P, _P, _P]),
    "cusrl_bce_pair_fwd_bwd": (c_int, [_P, c_int64, c_float, _P, _P, _P]),
    "cusrl_graph_census": (c_int, [_P, POINTER(c_int64), c_int, ctypes.c_char_p, c_int64, POINTER(c_int64)]),
    "cusrl_graph_replace_memsets": (c_int, [_P, POINTER(c_int64)]),
    "cusrl_comm_available": (c_int, []),
    "cusrl_comm_last_error": (c_char_p, []),
    "cusrl_comm_unique_id": (c_int, [_P]),
    "cusrl_comm_create": (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    "cusrl_comm_destroy": (c_int, [_P]),
    "cusrl_comm_abort": (c_int, [_P]),
    "cusrl_comm_world_size": (c_int, [_P]),
    "cusrl_allreduce_mean": (c_int, [_P, c_int64, _P, _P]),
    "cusrl_allgather": (c_int, [_P, _P, c_int64, _P, _P]),
    "cusrl_broadcast": (c_int, [_P, c_int64, c_int, _P, _P]),
    "cusrl_sequence_count": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P]),
    "cusrl_sequence_blocks": (c_int64, [c_int64]),
    "cusrl_sequence_layout": (c_int, [_P, c_int64, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "cusrl_gather_memory": (c_int, [_P, _P, _P, _P, c_int64, c_int64, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib() -> ctypes.CDLL:
    """Load the HIP library once; fail loudly when it is missing or stale."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError(
                f"{LIB_PATH} is missing: the gfx950 HIP extension has not been built. "
                "Run `python __graft_entry__.py` (or `__graft_entry__.build()`) first; "
                "cusrl_amd has no CPU / eager fallback for the rollout + PPO-update hot path."
            )
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = stale library
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.cusrl_abi_version() != ABI_VERSION:
            raise NativeError(f"ABI mismatch: library {handle.cusrl_abi_version()}, binding {ABI_VERSION}; rebuild")
        _lib = handle
        _options_from_environment()
    return _lib


# The A/B scripts of rounds 2-5 drive the kernels' launch-shape / cache-policy overrides through CUSRL_* environment variables.
# The library no longer reads the environment in its launch entry points (cusrl_set_option, ABI 6): the HOST translates them,
# once, when it loads the library.  {environment variable: (option, {text: value})}; None = the integer as it stands.
_ENVIRONMENT_OPTIONS = {
    "CUSRL_GAE_POLICY": ("gae_policy", {"0": 1, "5": 6, "7": 8}),
    "CUSRL_GAE_BLOCK": ("gae_block", None),
    "CUSRL_LOSS_POLICY": ("loss_policy", {"0": 1, "1": 2}),
    "CUSRL_PUSH_POLICY": ("push_policy", {"0": 1, "3": 2}),
    "CUSRL_COLSUM_ROWS": ("colsum_rows", None),
    "CUSRL_HEAD_ROWS": ("head_rows", None),
    "CUSRL_GRU_BIAS_ROWS": ("gru_bias_rows", None),
}


def set_option(key: str, value: int) -> None:
    """``cusrl_set_option``: force a kernel's launch shape / cache policy (0: back to its own rule); include/cusrl_hip.h."""
    check(lib().cusrl_set_option(key.encode(), int(value)), f"cusrl_set_option({key!r}, {value})")


def get_option(key: str) -> int:
    value = c_int64()
    check(lib().cusrl_get_option(key.encode(), ctypes.byref(value)), f"cusrl_get_option({key!r})")
    return int(value.value)


def _options_from_environment() -> None:
    for variable, (key, mapping) in _ENVIRONMENT_OPTIONS.items():
        text = os.environ.get(variable)
        if text is None or text == "":
            continue
        try:
            value = mapping[text] if mapping is not None else int(text)
        except (KeyError, ValueError):
            continue  # (an unknown value meant "the kernel's own rule" to the library, too)
        if _lib.cusrl_set_option(key.encode(), value) != 0:
            continue


# Launch census: how often each C-ABI entry point was called through ``check`` (every ``ops`` function reports its
# launch here).  Tests use it to prove that a result came from the HIP kernel and not from a torch-op form
# (``launch_counts["cusrl_rnd_reward"]`` must move when the RND hook runs on a GPU).
launch_counts: dict[str, int] = {}


def check(code: int, what: str) -> None:
    launch_counts[what] = launch_counts.get(what, 0) + 1
    if code != 0:
        text = lib().cusrl_error_string(code).decode()
        if code == -4:  # CUSRL_E_COMM
            text += ": " + lib().cusrl_comm_last_error().decode()
        raise NativeError(f"{what} failed with code {code}: {text}")
