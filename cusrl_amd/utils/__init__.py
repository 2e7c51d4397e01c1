from cusrl_amd.utils import distributed
from cusrl_amd.utils.config import CONFIG, configure_distributed, device, is_autocast_available
from cusrl_amd.utils.distributed import is_main_process
from cusrl_amd.utils.metrics import Metrics
from cusrl_amd.utils.misc import get_first, set_global_seed
from cusrl_amd.utils.timing import Timer

__all__ = [
    "CONFIG",
    "Metrics",
    "Timer",
    "configure_distributed",
    "device",
    "distributed",
    "get_first",
    "is_autocast_available",
    "is_main_process",
    "set_global_seed",
]
