from cusrl_amd.hook.on_policy.advantage import AdvantageNormalization, AdvantageReduction
from cusrl_amd.hook.on_policy.common import OnPolicyPreparation
from cusrl_amd.hook.on_policy.gae import GeneralizedAdvantageEstimation
from cusrl_amd.hook.on_policy.gradient_clipping import GradientClipping
from cusrl_amd.hook.on_policy.lr_schedule import AdaptiveLRSchedule, MiniBatchWiseLRSchedule, ThresholdLRSchedule
from cusrl_amd.hook.on_policy.ppo import EntropyLoss, PpoSurrogateLoss
from cusrl_amd.hook.on_policy.stats import OnPolicyStatistics
from cusrl_amd.hook.on_policy.value import ValueComputation, ValueLoss

__all__ = [
    "AdaptiveLRSchedule",
    "AdvantageNormalization",
    "AdvantageReduction",
    "EntropyLoss",
    "GeneralizedAdvantageEstimation",
    "GradientClipping",
    "MiniBatchWiseLRSchedule",
    "OnPolicyPreparation",
    "OnPolicyStatistics",
    "PpoSurrogateLoss",
    "ThresholdLRSchedule",
    "ValueComputation",
    "ValueLoss",
]
