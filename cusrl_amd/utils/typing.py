"""Type aliases shared across the package (mirrors the names of cusrl/utils/typing.py)."""

from collections.abc import Mapping, Sequence
from typing import TypeAlias, TypeVar, Union

import numpy as np
import torch

Array: TypeAlias = Union[np.ndarray, torch.Tensor]
ArrayT = TypeVar("ArrayT", np.ndarray, torch.Tensor)
Slice: TypeAlias = Union[slice, Sequence[int]]
_T = TypeVar("_T")
Nested: TypeAlias = Union[_T, list, tuple, Mapping]
NestedArray: TypeAlias = Nested
NestedTensor: TypeAlias = Nested
Memory: TypeAlias = Union[torch.Tensor, dict, None]
