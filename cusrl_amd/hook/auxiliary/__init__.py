from cusrl_amd.hook.auxiliary.amp import AdversarialMotionPrior
from cusrl_amd.hook.auxiliary.rnd import RandomNetworkDistillation

__all__ = ["AdversarialMotionPrior", "RandomNetworkDistillation"]
