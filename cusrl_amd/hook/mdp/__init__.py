from cusrl_amd.hook.mdp.observation import ObservationNormalization
from cusrl_amd.hook.mdp.reward import RewardShaping

__all__ = ["ObservationNormalization", "RewardShaping"]
