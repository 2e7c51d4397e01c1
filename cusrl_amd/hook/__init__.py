from cusrl_amd.hook.control import ModuleInitialization
from cusrl_amd.hook.mdp import ObservationNormalization
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
