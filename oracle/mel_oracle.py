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


def istft_frames(spec: torch.Tensor, window: torch.Tensor, hop: int) -> torch.Tensor:
    """Documented torch.istft(center=True, length=None) semantics: irfft per frame, * window, overlap-add, divide by the
    overlap-added squared window, trim n_fft/2 on both sides.  spec: (rows, n_fft/2+1, T) complex -> (rows, hop*(T-1))."""
    n = window.numel()
    rows, _, T = spec.shape
    frames = torch.fft.irfft(spec.transpose(-1, -2), n=n, dim=-1) * window                 # (rows, T, n)
    total = n + hop * (T - 1)
    y = torch.zeros(rows, total, dtype=window.dtype)
    env = torch.zeros(total, dtype=window.dtype)
    for t in range(T):
        y[:, t * hop:t * hop + n] += frames[:, t]
        env[t * hop:t * hop + n] += window ** 2
    return (y / env)[:, n // 2: n // 2 + hop * (T - 1)]


def unmel(mel_amp: torch.Tensor, filters: torch.Tensor) -> torch.Tensor:
    """frequency_scale.py:130-142: minimum-norm solution of filters^T X = mel (lstsq 'gels', full row rank), then relu."""
    shape = mel_amp.shape
    m = mel_amp.reshape(-1, shape[-2], shape[-1])
    x = torch.linalg.lstsq(filters.t()[None], m, driver="gels").solution
    return torch.relu(x).view(shape[:-2] + (filters.shape[0], shape[-1]))


def griffinlim(spec: torch.Tensor, window: torch.Tensor, hop: int, n_iter: int, momentum: float, stereo_coherence: float,
               dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """phase_recovery.py:39-129 (rand_init False, stereo True): spec (B, 2, n_stft, T) linear magnitudes -> (B, 2, L).
    `dtype=torch.float64` runs the same iteration in double precision: the iteration re-normalises near-empty bins to unit
    phasors, which makes the float32 result itself rounding-sensitive (~1 % after one iteration, ~8 % after four on the
    golden input), so implementations are judged by their distance to the float64 trajectory."""
    m = momentum / (1 + momentum)
    shape = spec.shape
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    window = window.to(dtype)
    s = spec.to(dtype).reshape(-1, shape[-2], shape[-1])
    merged = ((s[0::2] + s[1::2]) / 2).repeat_interleave(2, dim=0)
    angles = torch.ones(1, shape[-2], shape[-1], dtype=cdt)
    tprev = torch.zeros((), dtype=cdt)
    for i in range(n_iter):
        t = i / n_iter - stereo_coherence
        mags = merged + t * (s - merged) if t > 0 else merged
        wave = istft_frames(angles * mags, window, hop)
        rebuilt = stft_frames(wave, window, hop)
        # phase_recovery.py:110-119: `angles = rebuilt; angles.sub_(tprev, alpha=momentum)` is IN PLACE on the tensor that
        # `tprev = rebuilt` then keeps, so the carried state is u_i = rebuilt_i - m * u_{i-1}, not rebuilt_i itself
        tprev = rebuilt - m * tprev
        angles = tprev / (tprev.abs() + 1e-16)
    wave = istft_frames(angles * s, window, hop)
    return wave.reshape(shape[:-2] + wave.shape[-1:])


def mel_to_raw(samples: torch.Tensor, *, window: torch.Tensor, hop: int, filters: torch.Tensor, n_iter: int, exponent: float = 0.25,
               mean: float = 1.295, scale: float = 2.247, momentum: float = 0.99, stereo_coherence: float = 0.67,
               dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """spectrogram.py:228-238,181-185: undo the affine + exponent, un-mel, FGLA."""
    amp = (samples / scale + mean).clip(min=0) ** (1 / exponent)
    return griffinlim(unmel(amp, filters), window, hop, n_iter, momentum, stereo_coherence, dtype)


# ----------------------------------------------------------------------------- MS_MDCT_DualFormat.raw_to_mel_spec (a-10 sibling)

def blackman_harris_window(n: int, exponent: float) -> torch.Tensor:
    """utils/mclt.py:69-71 (periodic 4-term Blackman-Harris) ** exponent, formats/ms_mdct_dual.py:91-95."""
    x = torch.arange(n) / n * 2 * torch.pi
    return (0.35875 - 0.48829 * torch.cos(x) + 0.14128 * torch.cos(2 * x) - 0.01168 * torch.cos(3 * x)) ** exponent


def mel_density(hz: torch.Tensor) -> torch.Tensor:
    """frequency_scale.py:36-37."""
    return 1127.0 / (700.0 + hz)


def slaney_mel_filterbank(n_stft: int, n_mel: int, fmin: float, fmax: float, sample_rate: int) -> torch.Tensor:
    """frequency_scale.py:151-168 with filter_norm='slaney' (triangular)."""
    fb = mel_filterbank(n_stft, n_mel, fmin, fmax, sample_rate)
    mels = torch.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mel + 2)
    pts = 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    return fb * (2.0 / (pts[2:n_mel + 2] - pts[:n_mel])).unsqueeze(0)


def raw_to_ms_mel_spec(audio: torch.Tensor, *, n_fft: int = 4096, hop: int = 256, n_mel: int = 256, sample_rate: int = 32000,
                       exp_low: float = 17.0, exp_high: float = 58.0, abs_exponent: float = 1.0, scale: float = 50.0, offset: float = 0.0,
                       freq_min: float = 0.0) -> torch.Tensor:
    """formats/ms_mdct_dual.py:229-257 (the non-sliced branch; ms_freq_min = 0 makes _high_pass the identity, :187-192):
    two window-normalised magnitude STFTs (torchaudio Spectrogram(power=1, normalized='window') == |torch.stft| / sqrt(sum w^2)),
    blended per bin by (mel density / max)^2, divided by the mel density, slaney mel bank, ** exponent * scale + offset."""
    nb = n_fft // 2 + 1
    hz = torch.linspace(0, sample_rate / 2, nb)
    dens = mel_density(hz).view(1, 1, -1, 1)
    blend = ((mel_density(hz) / mel_density(hz).amax()) ** 2).view(1, 1, -1, 1)
    specs = []
    for e in (exp_low, exp_high):
        w = blackman_harris_window(n_fft, e)
        specs.append(stft_frames(audio.float(), w, hop).abs() / w.pow(2).sum().sqrt())
    blended = specs[0] * blend + specs[1] * (1 - blend)
    fb = slaney_mel_filterbank(nb, n_mel, freq_min, sample_rate / 2, sample_rate)
    mel = torch.matmul((blended / dens).transpose(-1, -2), fb).transpose(-1, -2)
    return mel ** abs_exponent * scale + offset


def ms_mel_to_mdct_psd(mel_spec: torch.Tensor, *, n_stft: int = 2049, n_mel: int = 256, sample_rate: int = 32000, freq_min: float = 0.0,
                       abs_exponent: float = 1.0, mel_offset: float = 0.0, scale: float = 0.18, offset: float = 0.0,
                       dtype: torch.dtype = torch.float64) -> torch.Tensor:
    """formats/ms_mdct_dual.py:259-271 with mdct_psd_num_bins == n_stft - 1: minimum-norm un-mel (frequency_scale.py:130-142,
    rectify=False) of the clipped spectrogram, last bin cropped, scaled.  Solved in float64 by default (the reference's float32
    `gels` solution is a few 1e-4 away from it)."""
    fb = slaney_mel_filterbank(n_stft, n_mel, freq_min, sample_rate / 2, sample_rate).to(dtype)
    x = (mel_spec.to(dtype) - mel_offset).clip(min=0) ** (1 / abs_exponent)
    shp = x.shape
    sol = torch.linalg.lstsq(fb.t()[None], x.reshape(-1, shp[-2], shp[-1]), driver="gelsd" if dtype == torch.float64 else "gels").solution
    return (sol.reshape(shp[:-2] + (n_stft, shp[-1]))[:, :, :-1, :] * scale + offset).float()
