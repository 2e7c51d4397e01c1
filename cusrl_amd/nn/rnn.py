"""Recurrent backbones (counterpart of cusrl/nn/module/rnn.py:21-449): ``nn.GRU`` / ``nn.LSTM`` / ``nn.RNN`` (on the GPU: rocBLAS
GEMMs + one HIP gate pass per time step, nn/gru.py; MIOpen only for what that path does not take) behind one wrapper that keeps memories as ``[N, layers * hidden]`` tensors (a dict ``{"hidden", "cell"}`` for LSTM),
steps one env step at a time during rollout and, on temporal minibatches, cuts the batch at episode boundaries so every
segment restarts from a zero memory (done-split layout from cusrl_amd/nn/recurrent.py — HIP kernels)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

from torch import Tensor, nn

from cusrl_amd.nn import recurrent
from cusrl_amd.nn.gru import gru_forward, gru_supported, lstm_forward, rnn_forward
from cusrl_amd.nn.module import Module, ModuleFactory
from cusrl_amd.utils.nest import map_nested

__all__ = ["Gru", "Lstm", "Rnn"]


def _to_layers(memory: Tensor, layers: int, hidden: int) -> Tensor:
    """``[N, layers * hidden]`` -> ``[layers, N, hidden]`` (what torch's RNNs take)."""
    return memory.reshape(memory.size(0), layers, hidden).transpose(0, 1).contiguous()


def _from_layers(state: Tensor) -> Tensor:
    return state.transpose(0, 1).reshape(state.size(1), -1)


class _Gru(nn.GRU):
    """``nn.GRU`` parameters; fp32 device batches run as GEMMs + one HIP gate pass per step (nn/gru.py) instead of MIOpen's
    RNN.  ``lengths`` (extension): per-sequence valid lengths on the device — the result of the packed form of the batch."""

    def forward(self, input, memory=None, lengths=None):
        if gru_supported(self, input):
            h0 = None if memory is None else _to_layers(memory, self.num_layers, self.hidden_size)
            output, hn = gru_forward(self, input, h0, lengths)
            return output, _from_layers(hn)
        if lengths is not None:
            raise ValueError("'lengths' needs the fused GRU path (fp32 device tensors); pass a PackedSequence instead")
        if memory is None:
            output, hn = super().forward(input)
        else:
            output, hn = super().forward(input, _to_layers(memory, self.num_layers, self.hidden_size))
        return output, _from_layers(hn)


class _VanillaRnn(nn.RNN):
    def forward(self, input, memory=None, lengths=None):
        if gru_supported(self, input):
            h0 = None if memory is None else _to_layers(memory, self.num_layers, self.hidden_size)
            output, hn = rnn_forward(self, input, h0, lengths)
            return output, _from_layers(hn)
        if lengths is not None:
            raise ValueError("'lengths' needs the fused RNN path (fp32 device tensors); pass a PackedSequence instead")
        if memory is None:
            output, hn = super().forward(input)
        else:
            output, hn = super().forward(input, _to_layers(memory, self.num_layers, self.hidden_size))
        return output, _from_layers(hn)


class _Lstm(nn.LSTM):
    """``nn.LSTM`` parameters; fp32 device batches run as GEMMs + one HIP gate pass per step (nn/gru.py)."""

    def forward(self, input, memory=None, lengths=None):
        if gru_supported(self, input):
            state = None if memory is None else (_to_layers(memory["hidden"], self.num_layers, self.hidden_size),
                                                 _to_layers(memory["cell"], self.num_layers, self.hidden_size))
            output, (hn, cn) = lstm_forward(self, input, state, lengths)
            return output, {"hidden": _from_layers(hn), "cell": _from_layers(cn)}
        if lengths is not None:
            raise ValueError("'lengths' needs the fused LSTM path (fp32 device tensors); pass a PackedSequence instead")
        if memory is None:
            output, (hn, cn) = super().forward(input)
        else:
            h0 = _to_layers(memory["hidden"], self.num_layers, self.hidden_size)
            c0 = _to_layers(memory["cell"], self.num_layers, self.hidden_size)
            output, (hn, cn) = super().forward(input, (h0, c0))
        return output, {"hidden": _from_layers(hn), "cell": _from_layers(cn)}


@dataclass
class RnnFactory(ModuleFactory):
    module_type: str
    hidden_size: int
    num_layers: int = 1
    nonlinearity: str = "tanh"
    bias: bool = True
    dropout: float = 0.0

    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        assert input_dim is not None
        kind = self.module_type.lower()
        common = dict(input_size=input_dim, hidden_size=self.hidden_size, num_layers=self.num_layers, bias=self.bias,
                      dropout=self.dropout)
        if kind in ("rnn", "vanilla"):
            core = _VanillaRnn(nonlinearity=self.nonlinearity, **common)
        elif kind == "lstm":
            core = _Lstm(**common)
        elif kind == "gru":
            core = _Gru(**common)
        else:
            raise ValueError(f"Unsupported RNN module class '{self.module_type}'")
        return Rnn(core, output_dim=output_dim)


class Rnn(Module):
    Factory = RnnFactory

    def __init__(self, rnn: nn.Module, output_dim: int | None = None):
        super().__init__(rnn.input_size, output_dim or rnn.hidden_size, is_recurrent=True)
        self.rnn = rnn
        self.output_proj = nn.Linear(rnn.hidden_size, output_dim) if output_dim else nn.Identity()

    def forward(self, input: Tensor, memory: Any = None, *, done: Tensor | None = None, sequential: bool = True,
                pack_sequence: bool = False, **kwargs):
        """``input`` is ``[L, N, C]`` (``sequential``) or ``[N, C]``; with ``done [L, N, 1]`` the memory restarts from zero
        after every finished episode inside the batch (rnn.py:206-253)."""
        if sequential and input.dim() >= 3:
            memory = recurrent.select_initial_memory(memory, input.shape[:-1])
        if done is not None:
            if not sequential:
                raise ValueError("'done' can be provided only when 'sequential' is True")
            latent, memory = self._forward_sequence(input, memory, done, pack_sequence=pack_sequence)
        else:
            latent, memory = self._forward_tensor(input, memory, sequential=sequential)
        return self.output_proj(latent), memory

    def _forward_tensor(self, input: Tensor, memory: Any = None, sequential: bool = True):
        shape = input.shape
        if input.dim() < 3:
            flat = input.reshape(1, -1, input.size(-1))
            if memory is not None:
                memory = map_nested(lambda m: m.reshape(flat.size(1), -1), memory)
        else:
            flat = input.reshape(input.size(0) if sequential else 1, -1, input.size(-1))
            if memory is not None:
                memory = map_nested(lambda m: m.flatten(0, -2), memory)
        latent, out_memory = self.rnn(flat, memory)
        latent = latent.reshape(*shape[:-1], latent.size(-1))
        if out_memory is not None:
            batch_dims = shape[(1 if sequential and len(shape) > 2 else 0):-1]
            out_memory = map_nested(lambda m: m.reshape(*batch_dims, m.size(-1)), out_memory)
        return latent, out_memory

    def _forward_sequence(self, input: Tensor, memory: Any, done: Tensor, pack_sequence: bool = False):
        layout = recurrent.compute_sequence_layout(done)
        padded_input, _ = recurrent.split_and_pad_sequences(input, done, layout)
        scattered = recurrent.scatter_memory(memory, done, layout)
        if pack_sequence:
            # rnn.py:273-291: the padded batch goes through the RNN as a PackedSequence, so every sequence stops at ITS
            # last valid step and the returned state is the true final state; gather_memory maps it back to envs.
            # (The PackedSequence API wants the lengths on the host: one read-back, as in the reference.)
            if input.dim() != 3:
                raise ValueError(f"Packed RNN input must be 3D, but got {input.dim()} dimensions")
            if isinstance(self.rnn, (_Gru, _Lstm, _VanillaRnn)) and gru_supported(self.rnn, padded_input):
                # same result, no packing and no host read of the lengths: the gate kernel stops every sequence at its own end
                padded_latent, scattered_output = self.rnn(padded_input, scattered, lengths=layout.lengths)
            else:
                packed = nn.utils.rnn.pack_padded_sequence(padded_input, lengths=layout.lengths.cpu(), enforce_sorted=False)
                packed_latent, scattered_output = self.rnn(packed, scattered)
                padded_latent, _ = nn.utils.rnn.pad_packed_sequence(packed_latent, total_length=padded_input.size(0))
            output_memory = recurrent.gather_memory(scattered_output, done, layout)
        elif padded_input.dim() == 3 and isinstance(self.rnn, (_Gru, _Lstm, _VanillaRnn)) and gru_supported(self.rnn, padded_input):
            # the fused cores skip the padded steps (length-sorted, shrinking per-step launches); what they would have
            # produced there is dropped by the unpad below anyway
            padded_latent, _ = self.rnn(padded_input, scattered, lengths=layout.lengths)
            output_memory = None
        else:
            padded_latent, _ = self._forward_tensor(padded_input, scattered)
            # the RNN also consumed padded steps, so its final state is not the state at each episode's last valid step
            output_memory = None
        return recurrent.unpad_and_merge_sequences(padded_latent, layout), output_memory

    def step_memory(self, input: Tensor, memory: Any = None, sequential: bool = True, **kwargs):
        if sequential and input.dim() >= 3:
            memory = recurrent.select_initial_memory(memory, input.shape[:-1])
        return self._forward_tensor(input, memory, sequential=sequential)[1]


class Gru(Rnn):
    def __init__(self, input_dim: int, hidden_size: int, num_layers: int = 1, bias: bool = True, dropout: float = 0.0,
                 output_dim: int | None = None):
        super().__init__(_Gru(input_dim, hidden_size, num_layers, bias=bias, dropout=dropout), output_dim=output_dim)


class Lstm(Rnn):
    def __init__(self, input_dim: int, hidden_size: int, num_layers: int = 1, bias: bool = True, dropout: float = 0.0,
                 output_dim: int | None = None):
        super().__init__(_Lstm(input_dim, hidden_size, num_layers, bias=bias, dropout=dropout), output_dim=output_dim)
