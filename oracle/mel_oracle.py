"""CPU oracle for the mel-STFT / FGLA format path  --  TEST INFRASTRUCTURE ONLY (see oracle/edm2_oracle.py header).

Parity status: PINNED by `tools/make_golden.py` against the reference's SpectrogramFormat
(/root/reference/src/modules/formats/old/spectrogram.py) run in the build container.  The STFT arithmetic of the reference
lives in third-party torchaudio (not vendored, version unpinned: environment.yml:13): `torchaudio.transforms.Spectrogram`
is a thin wrapper over `torch.stft`; its documented semantics are restated here with explicit framing + `torch.fft.rfft`
(never `torch.stft` itself, so the oracle is an independent statement), and anchored on the reference's call sites
(spectrogram.py:116-128,176-179) through the golden vectors.
"""
from __future__ import annotations

import math

import torch

from .edm2_oracle import hz_to_mel


def hann_power_window(n: int, exponent: float, periodic: bool = True) -> torch.Tensor:
    """spectrogram.py:99-104: hann(n, periodic) ** exponent (float32)."""
    return torch.hann_window(n, periodic=periodic) ** exponent


def mel_filterbank(n_stft: int, n_mel: int, fmin: float, fmax: float, sample_rate: int) -> torch.Tensor:
    """frequency_scale.py:45-58,151-168 (triangular, no norm): (n_stft, n_mel) float32, with the reference's dtype
    sequence (float64 endpoints -> float32 linspace -> float32 mel->Hz)."""
    bins = torch.linspace(0, sample_rate / 2, n_stft)
    mels = torch.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mel + 2)
    pts = 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    diff = pts[1:] - pts[:-1]
    slopes = pts.unsqueeze(0) - bins.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def stft_frames(audio: torch.Tensor, window: torch.Tensor, hop: int) -> torch.Tensor:
    """Documented torch.stft(center=True, pad_mode='reflect', onesided=True, normalized=False) semantics:
    frame t = window * x_reflect_padded[t*hop : t*hop + n_fft]; returns complex (..., n_fft/2+1, T)."""
    n = window.numel()
    x = audio.reshape(-1, audio.shape[-1])
    xp = torch.nn.functional.pad(x.unsqueeze(1), (n // 2, n // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n, hop) * window                      # (rows, T, n)
    spec = torch.fft.rfft(frames, dim=-1).transpose(-1, -2)      # (rows, n/2+1, T)
    return spec.reshape(audio.shape[:-1] + spec.shape[-2:])


def raw_to_mel(audio: torch.Tensor, *, window: torch.Tensor, hop: int, filters: torch.Tensor, exponent: float = 0.25,
               mean: float = 1.295, scale: float = 2.247) -> torch.Tensor:
    """spectrogram.py:176-179,217-226: (|STFT|^T @ filters)^T ** exponent, then (x - mean) * scale."""
    mag = stft_frames(audio, window, hop).abs()
    mel = torch.matmul(mag.transpose(-1, -2), filters).transpose(-1, -2)
    return (mel ** exponent - mean) * scale
