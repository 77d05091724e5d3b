"""Mel / log frequency-axis tables (host side, built once per configuration).

Counterpart of reference src/modules/formats/frequency_scale.py:85-168.  Only the *tables* live here (filter
bank values, band edges, warped frequency points); applying them to spectrogram frames is done by the HIP
mel kernels.  The dtype sequence of the reference is reproduced on purpose (float64 python endpoints ->
float32 `linspace` -> float32 mel->Hz): recomputing the bank in float64 moves one band edge
(SURVEY.md section 8 a-11), and the band edges are part of the bit-exact contract.
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def hz_to_mel(freq: float) -> float:
    return 2595.0 * math.log10(1.0 + (freq / 700.0))


def mel_to_hz(mels: torch.Tensor) -> torch.Tensor:
    return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)


class FrequencyScale:
    """Warped frequency axis with a triangular filter bank of `num_filters` bands over `num_stft_bins` bins."""

    def __init__(self, freq_scale: str = "mel", freq_min: float = 0.0, freq_max: Optional[float] = None, sample_rate: int = 32000,
                 num_stft_bins: int = 3201, num_filters: int = 256, filter_norm: Optional[str] = None) -> None:
        if freq_scale not in ("mel", "log"):
            raise ValueError(f"Unknown frequency scale: {freq_scale}")
        self.freq_scale, self.freq_min = freq_scale, freq_min
        self.freq_max = freq_max or sample_rate / 2
        self.sample_rate, self.num_stft_bins, self.num_filters, self.filter_norm = sample_rate, num_stft_bins, num_filters, filter_norm
        self._filters: Optional[torch.Tensor] = None

    def _warp(self, f: float) -> float:
        return hz_to_mel(f) if self.freq_scale == "mel" else math.log2(f)

    def get_unscaled(self, num_points: int, device=None) -> torch.Tensor:
        """`num_points` frequencies (Hz) equally spaced on the warped axis between freq_min and freq_max."""
        pts = torch.linspace(self._warp(self.freq_min), self._warp(self.freq_max), num_points, device=device)
        return mel_to_hz(pts) if self.freq_scale == "mel" else torch.exp2(pts)

    @property
    def filters(self) -> torch.Tensor:
        """(num_stft_bins, num_filters) float32 triangular bank, values as the reference computes them."""
        if self._filters is None:
            bins = torch.linspace(0, self.sample_rate / 2, self.num_stft_bins)
            pts = self.get_unscaled(self.num_filters + 2)
            width = pts[1:] - pts[:-1]
            dist = pts.unsqueeze(0) - bins.unsqueeze(1)
            rising = (-1.0 * dist[:, :-2]) / width[:-1]
            falling = dist[:, 2:] / width[1:]
            fb = torch.max(torch.zeros(1), torch.min(rising, falling))
            if self.filter_norm == "slaney":
                fb = fb * (2.0 / (pts[2:self.num_filters + 2] - pts[:self.num_filters])).unsqueeze(0)
            self._filters = fb
        return self._filters

    def band_edges(self) -> torch.Tensor:
        """(num_filters, 2) int32: first and last STFT bin with non-zero weight of every filter (inclusive)."""
        nz = self.filters > 0
        idx = torch.arange(self.num_stft_bins).unsqueeze(1)
        first = torch.where(nz, idx, self.num_stft_bins).min(dim=0).values
        last = torch.where(nz, idx, -1).max(dim=0).values
        return torch.stack([first, last], dim=1).to(torch.int32)
