"""mel-spectrogram audio format on the HIP kernels.

Drop-in for reference src/modules/formats/old/spectrogram.py (`SpectrogramFormat`, named `modules.formats.spectrogram`
by the default model_index.json): same config fields and method signatures.  `raw_to_sample` is ONE fused kernel
(ddx_mel_stft); constant tables (window, twiddles, mel bands) are built on the host exactly as the reference builds them
and uploaded once.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch

from ... import _lib as L
from ..._lib import DDXError, check, current_stream, lib, ptr
from .format import DualDiffusionFormat, DualDiffusionFormatConfig
from .frequency_scale import FrequencyScale


FGLA_FUSED_ITER = True     # analysis_i + synth_{i+1} in one launch per frame (ddx_fgla_iter); tests / tools flip it for the A/B


@dataclass
class SpectrogramFormatConfig(DualDiffusionFormatConfig):
    raw_to_sample_scale: float = 2.247
    sample_to_raw_scale: float = 0.445
    sample_mean: float = 1.295
    abs_exp1_scale: float = 0.008
    abs_exp1_mel_density: bool = False
    unscaled_psd_scale: float = 0.625
    unscaled_psd_mel_density: bool = False
    unscaled_psd_num_fft_bins: int = 3328
    unscaled_psd_rectify: bool = True
    abs_exponent: float = 0.25
    step_size_ms: int = 8
    window_duration_ms: int = 200
    padded_duration_ms: int = 200
    window_exponent: float = 32
    window_periodic: bool = True
    freq_scale_type: str = "mel"
    num_frequencies: int = 256
    min_frequency: int = 20
    max_frequency: int = 16000
    freq_scale_norm: Optional[str] = None
    num_fgla_iters: int = 200
    fgla_momentum: float = 0.99
    stereo_coherence: float = 0.67

    @property
    def stereo(self) -> bool:
        return self.sample_raw_channels == 2

    @property
    def padded_length(self) -> int:
        return int(self.padded_duration_ms / 1000.0 * self.sample_rate)

    @property
    def win_length(self) -> int:
        return int(self.window_duration_ms / 1000.0 * self.sample_rate)

    @property
    def hop_length(self) -> int:
        return int(self.step_size_ms / 1000.0 * self.sample_rate)

    @property
    def num_stft_bins(self) -> int:
        return self.padded_length // 2 + 1


def fft_twiddles(n: int) -> torch.Tensor:
    """(n, 2) float32 table of exp(-2 pi i k / n), evaluated in float64."""
    k = torch.arange(n, dtype=torch.float64) * (2.0 * math.pi / n)
    return torch.stack([torch.cos(k), -torch.sin(k)], dim=1).to(torch.float32).contiguous()


class SpectrogramFormat(DualDiffusionFormat):

    config_class = SpectrogramFormatConfig

    def __init__(self, config: SpectrogramFormatConfig) -> None:
        super().__init__()
        self.config = config
        if config.win_length != config.padded_length:
            raise DDXError("the HIP mel-STFT is built for win_length == n_fft (the reference default: 200 ms / 200 ms)")
        self.freq_scale = FrequencyScale(config.freq_scale_type, config.min_frequency, config.max_frequency, config.sample_rate,
                                         config.num_stft_bins, config.num_frequencies, config.freq_scale_norm)
        # window exactly as the reference: hann(win, periodic) ** exponent in float32 (spectrogram.py:99-104)
        self.register_buffer("window", torch.hann_window(config.win_length, periodic=config.window_periodic) ** config.window_exponent,
                             persistent=False)
        self.register_buffer("twiddle", fft_twiddles(config.padded_length), persistent=False)
        edges = self.freq_scale.band_edges()
        fb = self.freq_scale.filters
        start, length = edges[:, 0].clone(), (edges[:, 1] - edges[:, 0] + 1)
        empty = length <= 0
        start[empty], length[empty] = 0, 0
        stride = int((int(length.max()) + 3) // 4 * 4)
        bw = torch.zeros(config.num_frequencies, stride)
        for m in range(config.num_frequencies):
            n = int(length[m])
            bw[m, :n] = fb[int(start[m]):int(start[m]) + n, m]
        self.register_buffer("band_start", start.to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("band_len", length.to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("band_w", bw.contiguous(), persistent=False)

    # ---- geometry (reference spectrogram.py:164-174, 203-215)
    def _spec_frames(self, audio_len: int) -> int:
        c = self.config
        return 1 + (audio_len + c.padded_length - c.win_length) // c.hop_length

    def sample_raw_crop_width(self, length: Optional[int] = None) -> int:
        c = self.config
        frames = self._spec_frames(length or c.sample_raw_length) // 128 * 128
        return (frames - 1) * c.hop_length + c.win_length - c.padded_length

    def get_sample_shape(self, bsz: int = 1, length: Optional[int] = None) -> tuple:
        c = self.config
        return (bsz, c.sample_raw_channels, c.num_frequencies, self._spec_frames(self.sample_raw_crop_width(length)))

    @torch.no_grad()
    def get_ln_freqs(self, x: torch.Tensor) -> torch.Tensor:
        """reference spectrogram.py:240-244 (host table; the modules upload the per-row values once per shape)."""
        ln = self.freq_scale.get_unscaled(x.shape[2] + 2, device="cpu")[1:-1].log2()
        ln = ln.view(1, 1, -1, 1).repeat(x.shape[0], 1, 1, x.shape[3])
        return ((ln - ln.mean()) / ln.std()).to(x.dtype)

    @property
    def ms_freq_scale(self) -> FrequencyScale:   # what UNet.get_ln_freqs reads (unet_edm2_b4.py:246)
        return self.freq_scale

    # ---- encode
    @torch.no_grad()
    def raw_to_sample(self, raw_samples: torch.Tensor) -> torch.Tensor:
        """(B, C, L) audio -> (B, C, n_mel, T) mel-spectrogram samples (reference spectrogram.py:217-226)."""
        if self.device.type != "cuda":
            raise DDXError("SpectrogramFormat is not on a ROCm device: no CPU fallback")
        c = self.config
        x = raw_samples.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cn, Ln = x.shape
        T = self._spec_frames(Ln)
        out = torch.empty(B, Cn, c.num_frequencies, T, device=self.device, dtype=torch.float32)
        d = L.MelStftDesc(audio=ptr(x), window=ptr(self.window), twiddle=ptr(self.twiddle), band_start=ptr(self.band_start),
                          band_len=ptr(self.band_len), band_w=ptr(self.band_w), out=ptr(out), B=B, C=Cn, L=Ln, T=T,
                          n_fft=c.padded_length, hop=c.hop_length, n_mel=c.num_frequencies, band_stride=self.band_w.shape[1],
                          exponent=c.abs_exponent, mean=c.sample_mean, scale=c.raw_to_sample_scale)
        check(lib().ddx_mel_stft(C.byref(d), current_stream()), "mel_stft")
        return out

    # ---- decode
    def _unmel_weights(self):
        """Minimum-norm un-mel operator: the reference solves lstsq(filters^T, mel) per call (frequency_scale.py:130-142);
        the operator is constant, so its pseudo-inverse is built once (float64) and applied as a 1x1 conv / GEMM."""
        if getattr(self, "_unmel", None) is None:
            from ... import ops
            pinv = torch.linalg.pinv(self.freq_scale.filters.double().t())          # (n_stft, n_mel)
            nb = self.config.num_stft_bins
            w = torch.zeros((nb + 3) // 4 * 4, self.config.num_frequencies, 1, 1)
            w[:nb, :, 0, 0] = pinv.float()
            wd = w.to(self.device).contiguous()
            # gain = sqrt(fan_in) cancels the 1/sqrt(fan_in) of the MPConv weight path exactly (256 -> 16/16)
            self._unmel = ops.wprep(wd, 1, torch.float32, gain=math.sqrt(self.config.num_frequencies), npix=0)
            self._unmel_keep = wd
        return self._unmel

    @torch.no_grad()
    def sample_to_raw(self, samples: torch.Tensor, n_fgla_iters: Optional[int] = None, quiet: bool = False) -> torch.Tensor:
        """(B, C, n_mel, T) mel samples -> (B, C, hop*(T-1)) audio by un-mel + FGLA (reference spectrogram.py:228-238)."""
        from ... import ops
        if self.device.type != "cuda":
            raise DDXError("SpectrogramFormat is not on a ROCm device: no CPU fallback")
        c = self.config
        dev, f32 = self.device, torch.float32
        x = samples.to(device=dev, dtype=f32).contiguous()
        B, Cn, n_mel, T = x.shape
        n_iter = n_fgla_iters or c.num_fgla_iters
        N, hop, nb = c.padded_length, c.hop_length, c.num_stft_bins
        st = current_stream()
        amp = torch.empty(B * Cn, 1, T, n_mel, device=dev, dtype=f32)
        check(lib().ddx_mel_to_amplitude(ptr(x), ptr(amp), B * Cn, n_mel, T, c.raw_to_sample_scale, c.sample_mean,
                                         1.0 / c.abs_exponent, st), "mel_to_amplitude")
        mags = ops.conv2d(amp, self._unmel_weights())                          # [B*C][1][T][nb padded to 4], relu applied on read
        mstride = mags.shape[3]
        us = (nb + 2) // 2 * 2                                     # state rows padded to an even bin count: 16-byte accesses
        u = torch.zeros(B, T, Cn, us, 2, device=dev, dtype=f32)   # state: u_i = rebuilt_i - m*u_{i-1} (see fgla.hip)
        frames = torch.empty(B, T, Cn, N, device=dev, dtype=f32)
        Lout = hop * (T - 1)
        audio = torch.empty(B, Cn, Lout, device=dev, dtype=f32)
        momentum = c.fgla_momentum / (1 + c.fgla_momentum)
        W, TW = ptr(self.window), ptr(self.twiddle)

        def synth(state, t_lerp, final):
            check(lib().ddx_fgla_synth(ptr(state), us, ptr(mags), W, TW, ptr(frames), B, Cn, T, N, mstride, t_lerp, int(final), st), "fgla_synth")
            check(lib().ddx_fgla_ola(ptr(frames), W, ptr(audio), B, Cn, T, N, hop, st), "fgla_ola")

        # reference loop: [synth_i, analysis_i] for i < n_iter, then the final synth (waveform = istft(angles * specgram)).  The analysis of
        # iteration i and the synthesis of iteration i + 1 (the final one after the last) are per-frame passes over the same state row:
        # ddx_fgla_iter runs them in one launch (the state is read once and written once per iteration; FGLA is HBM-bound)
        if not FGLA_FUSED_ITER:                                                     # (the reference's loop as three launches per iteration)
            for i in range(n_iter):
                synth(None if i == 0 else u, i / n_iter - c.stereo_coherence, False)
                check(lib().ddx_fgla_analysis(ptr(audio), W, TW, ptr(u), us, B, Cn, T, Lout, N, hop, momentum, st), "fgla_analysis")
            synth(u, 0.0, True)
            return audio
        synth(None, -c.stereo_coherence, False)                                     # i == 0: angles = 1 (rand_init False)
        for i in range(n_iter):
            last = i == n_iter - 1
            t_lerp = 0.0 if last else (i + 1) / n_iter - c.stereo_coherence
            check(lib().ddx_fgla_iter(ptr(audio), W, TW, ptr(u), us, ptr(mags), mstride, ptr(frames), B, Cn, T, Lout, N, hop, momentum,
                                      t_lerp, int(last), st), "fgla_iter")
            check(lib().ddx_fgla_ola(ptr(frames), W, ptr(audio), B, Cn, T, N, hop, st), "fgla_ola")
        return audio
