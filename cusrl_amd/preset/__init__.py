from cusrl_amd.preset.ppo import AdamFactory, PpoAgentFactory, RecurrentPpoAgentFactory, ppo_hook_suite

__all__ = ["AdamFactory", "PpoAgentFactory", "RecurrentPpoAgentFactory", "ppo_hook_suite"]
