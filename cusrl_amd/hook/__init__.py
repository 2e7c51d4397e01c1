from cusrl_amd.hook.control import ModuleInitialization
from cusrl_amd.hook.auxiliary import AdversarialMotionPrior, RandomNetworkDistillation
from cusrl_amd.hook.mdp import ObservationNormalization, RewardShaping
from cusrl_amd.hook.on_policy import (
    AdvantageNormalization,
    AdvantageReduction,
    EntropyLoss,
    GeneralizedAdvantageEstimation,
    GradientClipping,
    OnPolicyPreparation,
    OnPolicyStatistics,
    PpoSurrogateLoss,
    ValueComputation,
    ValueLoss,
)

__all__ = [
    "AdversarialMotionPrior",
    "RandomNetworkDistillation",
    "RewardShaping",
    "AdvantageNormalization",
    "AdvantageReduction",
    "EntropyLoss",
    "GeneralizedAdvantageEstimation",
    "GradientClipping",
    "ModuleInitialization",
    "ObservationNormalization",
    "OnPolicyPreparation",
    "OnPolicyStatistics",
    "PpoSurrogateLoss",
    "ValueComputation",
    "ValueLoss",
]
