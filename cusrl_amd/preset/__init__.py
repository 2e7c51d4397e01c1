from cusrl_amd.preset.ppo import AdamFactory, AmpAgentFactory, PpoAgentFactory, RecurrentPpoAgentFactory, ppo_hook_suite

__all__ = ["AdamFactory", "AmpAgentFactory", "PpoAgentFactory", "RecurrentPpoAgentFactory", "ppo_hook_suite"]
