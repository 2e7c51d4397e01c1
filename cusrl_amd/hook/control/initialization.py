"""Orthogonal initialisation of actor and critic (counterpart of cusrl/hook/control/initialization.py:12-125):
gain sqrt(2) for Linear / RNN weights, 0.1*sqrt(2) for the policy mean head, zero biases."""

from __future__ import annotations

import math

from torch import nn

from cusrl_amd.template.hook import Hook

__all__ = ["ModuleInitialization"]


class ModuleInitialization(Hook):
    def __init__(self, scale: float = math.sqrt(2), scale_dist: float = math.sqrt(2) * 0.1, zero_bias: bool = True,
                 init_actor: bool = True, init_critic: bool = True):
        super().__init__()
        self.scale, self.scale_dist, self.zero_bias = scale, scale_dist, zero_bias
        self.init_actor, self.init_critic = init_actor, init_critic

    def init(self):
        if self.init_actor:
            self._init_tree(self.agent.actor)
            if self.scale_dist != self.scale:
                self._init_linear(self.agent.actor.distribution.mean_head, self.scale_dist)
        if self.init_critic:
            self._init_tree(self.agent.critic)

    def _init_tree(self, root: nn.Module):
        for module in root.modules():  # same traversal order as the reference, so RNG consumption matches
            if isinstance(module, nn.Linear):
                self._init_linear(module, self.scale)
            elif isinstance(module, (nn.RNN, nn.LSTM, nn.GRU)):
                for layer in range(module.num_layers):
                    nn.init.orthogonal_(getattr(module, f"weight_hh_l{layer}"), gain=self.scale)
                    nn.init.orthogonal_(getattr(module, f"weight_ih_l{layer}"), gain=self.scale)
                    if self.zero_bias:
                        for bias in (f"bias_hh_l{layer}", f"bias_ih_l{layer}"):
                            if getattr(module, bias, None) is not None:
                                nn.init.zeros_(getattr(module, bias))

    def _init_linear(self, module: nn.Linear, gain: float):
        nn.init.orthogonal_(module.weight, gain=gain)
        if self.zero_bias and module.bias is not None:
            nn.init.zeros_(module.bias)
