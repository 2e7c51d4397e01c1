from cusrl_amd.hook.control import EmptyCudaCache, ModuleInitialization
from cusrl_amd.hook.auxiliary import AdversarialMotionPrior, RandomNetworkDistillation
from cusrl_amd.hook.mdp import ObservationNormalization, RewardShaping
from cusrl_amd.hook.on_policy import (
    AdaptiveLRSchedule,
    AdvantageNormalization,
    AdvantageReduction,
    EntropyLoss,
    GeneralizedAdvantageEstimation,
    GradientClipping,
    MiniBatchWiseLRSchedule,
    OnPolicyPreparation,
    OnPolicyStatistics,
    PpoSurrogateLoss,
    ThresholdLRSchedule,
    ValueComputation,
    ValueLoss,
)

__all__ = [
    "AdaptiveLRSchedule",
    "ThresholdLRSchedule",
    "AdversarialMotionPrior",
    "RandomNetworkDistillation",
    "RewardShaping",
    "AdvantageNormalization",
    "AdvantageReduction",
    "EmptyCudaCache",
    "EntropyLoss",
    "GeneralizedAdvantageEstimation",
    "GradientClipping",
    "MiniBatchWiseLRSchedule",
    "ModuleInitialization",
    "ObservationNormalization",
    "OnPolicyPreparation",
    "OnPolicyStatistics",
    "PpoSurrogateLoss",
    "ValueComputation",
    "ValueLoss",
]
