from cusrl_amd.template.actor_critic import ActorCritic, ActorCriticFactory
from cusrl_amd.template.agent import Agent, AgentFactory
from cusrl_amd.template.buffer import Buffer, Sampler
from cusrl_amd.template.environment import Environment, EnvironmentSpec
from cusrl_amd.template.hook import Hook, HookComposite
from cusrl_amd.template.optimizer import OptimizerCollection, OptimizerFactory, build_optimizer
from cusrl_amd.template.trainer import Trainer, TrainerHook

__all__ = [
    "ActorCritic",
    "ActorCriticFactory",
    "Agent",
    "AgentFactory",
    "Buffer",
    "Environment",
    "EnvironmentSpec",
    "Hook",
    "HookComposite",
    "OptimizerCollection",
    "OptimizerFactory",
    "Sampler",
    "Trainer",
    "TrainerHook",
    "build_optimizer",
]
