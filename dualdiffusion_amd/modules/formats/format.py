"""Abstract audio format (mirrors reference src/modules/formats/format.py:29-60)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional

import torch

from ..module import DualDiffusionModule, DualDiffusionModuleConfig


@dataclass
class DualDiffusionFormatConfig(DualDiffusionModuleConfig):
    sample_rate: int = 32000
    num_raw_channels: int = 2
    default_raw_length: int = 1408768
    # fields the shipped format.json still carries (config/models/default/format.json)
    sample_raw_channels: int = 2
    sample_raw_length: int = 1440000
    noise_floor: float = 2e-5
    t_scale: Optional[float] = None


class DualDiffusionFormat(DualDiffusionModule, ABC):
    module_name: str = "format"
    has_trainable_parameters: bool = False
    supports_half_precision: bool = False
    supports_compile: bool = False

    @abstractmethod
    def raw_to_sample(self, raw_samples: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def sample_to_raw(self, samples: torch.Tensor) -> torch.Tensor: ...
