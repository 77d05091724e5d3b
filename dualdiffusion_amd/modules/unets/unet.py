"""Abstract UNet interface (mirrors reference src/modules/unets/unet.py:31-65)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ..module import DualDiffusionModule, DualDiffusionModuleConfig


@dataclass
class DualDiffusionUNetConfig(DualDiffusionModuleConfig, ABC):
    in_channels: int = 4
    out_channels: int = 4
    in_channels_emb: int = 512
    dropout: float = 0.
    sigma_max: float = 200.
    sigma_min: float = 0.03
    sigma_data: float = 1.


class DualDiffusionUNet(DualDiffusionModule, ABC):
    module_name: str = "unet"

    @abstractmethod
    def get_embeddings(self, emb_in: torch.Tensor, conditioning_mask: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def get_sigma_loss_logvar(self, sigma: Optional[torch.Tensor] = None) -> torch.Tensor: ...

    @abstractmethod
    def get_latent_shape(self, latent_shape: Union[torch.Size, tuple]) -> torch.Size: ...

    @abstractmethod
    def forward(self, x_in: torch.Tensor, sigma: torch.Tensor, format, embeddings: Optional[torch.Tensor] = None,
                x_ref: Optional[torch.Tensor] = None) -> torch.Tensor: ...
