"""Abstract VAE interface and latent distributions (mirrors reference src/modules/old/vaes/vae.py:34-151)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ..module import DualDiffusionModule, DualDiffusionModuleConfig


class LatentsDistribution(ABC):
    @abstractmethod
    def sample(self) -> torch.Tensor: ...

    @abstractmethod
    def mode(self) -> torch.Tensor: ...

    @abstractmethod
    def kl(self, other: Optional["LatentsDistribution"] = None) -> torch.Tensor: ...


class IsotropicGaussianDistribution(LatentsDistribution):
    """Mean from the encoder, one constant log-variance (fixed target SNR): reference vae.py:48-82."""

    def __init__(self, parameters: torch.Tensor, logvar: torch.Tensor, deterministic: bool = False) -> None:
        self.deterministic = deterministic
        self.parameters = self.mean = parameters
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)
        else:
            self.std, self.var = torch.exp(0.5 * self.logvar), torch.exp(self.logvar)

    def sample(self) -> torch.Tensor:
        return self.mean + self.std * torch.randn_like(self.mean)

    def mode(self) -> torch.Tensor:
        return self.mean

    def kl(self, other: Optional["IsotropicGaussianDistribution"] = None) -> torch.Tensor:
        if self.deterministic:
            return torch.zeros(1, device=self.mean.device, dtype=self.mean.dtype)
        dims = tuple(range(self.mean.ndim))
        if other is None:
            return 0.5 * torch.mean(self.mean.square() + self.var - 1. - self.logvar, dim=dims)
        return 0.5 * torch.mean((self.mean - other.mean).square() / other.var + self.var / other.var - 1. - self.logvar + other.logvar, dim=dims)


@dataclass
class DualDiffusionVAEConfig(DualDiffusionModuleConfig, ABC):
    in_channels: int = 2
    in_num_freqs: int = 256
    in_channels_emb: int = 512
    out_channels: int = 2
    latent_channels: int = 4
    dropout: float = 0.
    latents_img_channel_order: Optional[tuple] = None


class DualDiffusionVAE(DualDiffusionModule, ABC):
    module_name: str = "vae"

    @abstractmethod
    def get_embeddings(self, emb_in: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def get_recon_loss_logvar(self) -> torch.Tensor: ...

    @abstractmethod
    def get_latent_shape(self, sample_shape: Union[torch.Size, tuple]) -> torch.Size: ...

    @abstractmethod
    def get_sample_shape(self, latent_shape: Union[torch.Size, tuple]) -> torch.Size: ...

    @abstractmethod
    def encode(self, x: torch.Tensor, embeddings: torch.Tensor, format) -> LatentsDistribution: ...

    @abstractmethod
    def decode(self, x: torch.Tensor, embeddings: torch.Tensor, format) -> torch.Tensor: ...
