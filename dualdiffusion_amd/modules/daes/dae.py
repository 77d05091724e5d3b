"""Abstract autoencoder of the live pipeline (mirrors reference src/modules/daes/dae.py:58-110)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional

import torch

from ..module import DualDiffusionModule, DualDiffusionModuleConfig


@dataclass
class DualDiffusionDAEConfig(DualDiffusionModuleConfig, ABC):
    in_channels: int = 2
    in_channels_emb: int = 1024
    in_num_freqs: int = 256
    out_channels: int = 2
    latent_channels: int = 4
    latents_img_split_stereo: bool = True
    latents_img_use_pca: bool = True
    latents_img_channel_order: Optional[tuple] = (1, 3, 2, 0)
    latents_img_flip_stereo: bool = False


class DualDiffusionDAE(DualDiffusionModule, ABC):
    module_name: str = "dae"

    @abstractmethod
    def get_embeddings(self, emb_in: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def get_recon_loss_logvar(self) -> torch.Tensor: ...

    @abstractmethod
    def get_latent_shape(self, sample_shape) -> torch.Size: ...

    @abstractmethod
    def get_mel_spec_shape(self, latent_shape) -> torch.Size: ...

    @abstractmethod
    def encode(self, x: torch.Tensor, embeddings: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def decode(self, x: torch.Tensor, embeddings: torch.Tensor) -> torch.Tensor: ...
