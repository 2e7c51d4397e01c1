from cusrl_amd.preset.ppo import AdamFactory, PpoAgentFactory, ppo_hook_suite

__all__ = ["AdamFactory", "PpoAgentFactory", "ppo_hook_suite"]
