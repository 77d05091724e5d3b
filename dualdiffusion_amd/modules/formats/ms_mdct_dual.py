"""Mel-scale spectrogram side of the live dual format on the HIP kernels.

Drop-in for the mel-spectrogram methods of reference src/modules/formats/ms_mdct_dual.py (`MS_MDCT_DualFormat`): same config
dataclass fields, `ms_freq_scale` (what `UNet.get_ln_freqs` reads, unet_edm2_b4.py:246), `get_raw_crop_width`,
`get_mel_spec_shape`, `raw_to_mel_spec` (:229-257) and `mel_spec_to_mdct_psd` (:259-271, the conditioning of the MCLT diffusion
decoder).  `raw_to_mel_spec` is ONE fused kernel (ddx_ms_mel_spec: two in-LDS FFT-4096 per frame, blend, banded slaney mel bank);
the constant tables are built on the host with the reference's dtype sequence and uploaded once.
The MDCT methods (`raw_to_mdct` / `mdct_to_raw`, utils/mclt.py) are outside the hot path of SURVEY.md section 8 and raise.
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
from .spectrogram import fft_twiddles


@dataclass
class MS_MDCT_DualFormatConfig(DualDiffusionFormatConfig):
    sample_rate: int = 32000
    num_raw_channels: int = 2
    default_raw_length: int = 1408768
    raw_to_mel_spec_scale: float = 50
    raw_to_mel_spec_offset: float = 0
    mel_spec_to_mdct_psd_scale: float = 0.18
    mel_spec_to_mdct_psd_offset: float = 0
    mdct_to_raw_scale: float = 2
    raw_to_mdct_scale: float = 12.1
    mdct_window_len: int = 512
    mdct_window_func: str = "kaiser_bessel_derived"
    mdct_psd_num_bins: int = 2048
    mdct_dual_channel: bool = False
    ms_abs_exponent: float = 1
    ms_filter_shape: str = "triangular"
    ms_freq_min: float = 0
    ms_width_alignment: int = 128
    ms_num_frequencies: int = 256
    ms_step_size_ms: int = 8
    ms_window_duration_ms: int = 128
    ms_padded_duration_ms: int = 128
    ms_window_exponent_low: float = 17
    ms_window_exponent_high: Optional[float] = 58
    ms_window_periodic: bool = True
    ms_window_func: str = "blackman_harris"

    @property
    def mdct_num_frequencies(self) -> int:
        return self.mdct_window_len // 2

    @property
    def ms_num_stft_bins(self) -> int:
        return self.ms_frame_padded_length // 2 + 1

    @property
    def ms_frame_padded_length(self) -> int:
        return int(self.ms_padded_duration_ms / 1000. * self.sample_rate)

    @property
    def ms_win_length(self) -> int:
        return int(self.ms_window_duration_ms / 1000. * self.sample_rate)

    @property
    def ms_frame_hop_length(self) -> int:
        return int(self.ms_step_size_ms / 1000. * self.sample_rate)


def get_mel_density(hz: torch.Tensor) -> torch.Tensor:
    """reference frequency_scale.py:36-37."""
    return 1127. / (700. + hz)


def _window(n: int, func: str, exponent: float, periodic: bool) -> torch.Tensor:
    """reference ms_mdct_dual.py:86-98 (float32 arithmetic like the reference's)."""
    if func == "blackman_harris":
        x = torch.arange(n) / n * 2 * torch.pi
        return (0.35875 - 0.48829 * torch.cos(x) + 0.14128 * torch.cos(2 * x) - 0.01168 * torch.cos(3 * x)) ** exponent
    if func == "hann":
        return torch.hann_window(n, periodic=periodic) ** exponent
    raise ValueError(f"Unsupported window function: {func}. Supported functions are 'hann' and 'blackman_harris'.")


class MS_MDCT_DualFormat(DualDiffusionFormat):

    config_class = MS_MDCT_DualFormatConfig

    def __init__(self, config: MS_MDCT_DualFormatConfig) -> None:
        super().__init__()
        self.config = c = config
        if c.ms_filter_shape != "triangular":
            raise NotImplementedError("MS_MDCT_DualFormat on the HIP path: only the triangular mel bank is built")
        if c.ms_win_length != c.ms_frame_padded_length or c.ms_frame_padded_length != 4096:
            raise DDXError("the HIP dual-window mel kernel is built for win_length == n_fft == 4096 (the reference default: 128 ms at 32 kHz)")
        nb = c.ms_num_stft_bins
        self.ms_freq_scale = FrequencyScale("mel", c.ms_freq_min, c.sample_rate / 2, c.sample_rate, nb, c.ms_num_frequencies, "slaney")
        ms_filter_freqs = self.ms_freq_scale.get_unscaled(c.ms_num_frequencies + 2)
        self.ms_lowest_filter_freq = float(ms_filter_freqs[1])
        hz = torch.linspace(0, c.sample_rate / 2, nb)
        dens = get_mel_density(hz)
        # windows pre-divided by their L2 norm (torchaudio Spectrogram(normalized="window"), ms_mdct_dual.py:110-139)
        w_low = _window(c.ms_win_length, c.ms_window_func, c.ms_window_exponent_low, c.ms_window_periodic)
        w_low = w_low / w_low.pow(2).sum().sqrt()
        if c.ms_window_exponent_high is not None:
            w_high = _window(c.ms_win_length, c.ms_window_func, c.ms_window_exponent_high, c.ms_window_periodic)
            w_high = w_high / w_high.pow(2).sum().sqrt()
            blend = (dens / dens.amax()) ** 2                        # :167-171
        else:
            w_high, blend = w_low, torch.ones(nb)
        self.register_buffer("window_low", w_low.contiguous(), persistent=False)
        self.register_buffer("window_high", w_high.contiguous(), persistent=False)
        self.register_buffer("bin_scale_low", (blend / dens).contiguous(), persistent=False)
        self.register_buffer("bin_scale_high", ((1 - blend) / dens).contiguous(), persistent=False)
        self.register_buffer("twiddle", fft_twiddles(c.ms_frame_padded_length), persistent=False)
        edges, fb = self.ms_freq_scale.band_edges(), self.ms_freq_scale.filters
        start, length = edges[:, 0].clone(), (edges[:, 1] - edges[:, 0] + 1)
        empty = length <= 0
        start[empty], length[empty] = 0, 0
        stride = int((int(length.max()) + 3) // 4 * 4)
        bw = torch.zeros(c.ms_num_frequencies, stride)
        for m in range(c.ms_num_frequencies):
            n = int(length[m])
            bw[m, :n] = fb[int(start[m]):int(start[m]) + n, m]
        self.register_buffer("band_start", start.to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("band_len", length.to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("band_w", bw.contiguous(), persistent=False)
        self._unmel = None

    # ---- geometry (reference :207-227)
    def _get_ms_shape(self, raw_shape: tuple) -> tuple:
        c = self.config
        num_frames = 1 + (raw_shape[-1] + c.ms_frame_padded_length - c.ms_win_length) // c.ms_frame_hop_length
        return tuple(raw_shape[:-1]) + (c.ms_num_frequencies, num_frames)

    def _get_ms_raw_shape(self, mel_spec_shape: tuple) -> tuple:
        c = self.config
        audio_len = (mel_spec_shape[-1] - 1) * c.ms_frame_hop_length + c.ms_win_length - c.ms_frame_padded_length
        return tuple(mel_spec_shape[:-2]) + (audio_len,)

    def get_raw_crop_width(self, raw_length: Optional[int] = None) -> int:
        c = self.config
        raw_length = raw_length or c.default_raw_length
        mel_spec_len = self._get_ms_shape((1, raw_length))[-1] // c.ms_width_alignment * c.ms_width_alignment
        return self._get_ms_raw_shape((1, mel_spec_len))[-1]

    def get_mel_spec_shape(self, bsz: int = 1, raw_length: Optional[int] = None) -> tuple:
        return self._get_ms_shape((bsz, self.config.num_raw_channels, self.get_raw_crop_width(raw_length)))

    # ---- encode (reference :229-257)
    @torch.no_grad()
    def raw_to_mel_spec(self, raw_samples: torch.Tensor, use_slicing: bool = False) -> torch.Tensor:
        """(B, C, L) audio -> (B, C, 256, T) mel-scale spectrogram.  `use_slicing` (the reference's per-sample memory saver) is
        accepted and irrelevant: nothing but audio and mel values touches HBM here."""
        if self.device.type != "cuda":
            raise DDXError("MS_MDCT_DualFormat is not on a ROCm device: no CPU fallback")
        c = self.config
        if c.ms_freq_min > 0 and (self.ms_lowest_filter_freq - c.ms_freq_min) > 0:
            raise NotImplementedError("ms_freq_min > 0 (the FFT high-pass of ms_mdct_dual.py:177-205) is not implemented on the HIP path")
        x = raw_samples.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cn, Ln = x.shape
        T = self._get_ms_shape((1, Ln))[-1]
        out = torch.empty(B, Cn, c.ms_num_frequencies, T, device=self.device, dtype=torch.float32)
        d = L.MsMelDesc(audio=ptr(x), window_low=ptr(self.window_low), window_high=ptr(self.window_high), twiddle=ptr(self.twiddle),
                        bin_scale_low=ptr(self.bin_scale_low), bin_scale_high=ptr(self.bin_scale_high), band_start=ptr(self.band_start),
                        band_len=ptr(self.band_len), band_w=ptr(self.band_w), out=ptr(out), B=B, C=Cn, L=Ln, T=T,
                        n_fft=c.ms_frame_padded_length, hop=c.ms_frame_hop_length, n_mel=c.ms_num_frequencies, band_stride=self.band_w.shape[1],
                        exponent=float(c.ms_abs_exponent), scale=float(c.raw_to_mel_spec_scale), offset=float(c.raw_to_mel_spec_offset))
        check(lib().ddx_ms_mel_spec(C.byref(d), current_stream()), "ms_mel_spec")
        return out

    # `raw_to_sample` of the abstract format = the mel spectrogram (what the autoencoder consumes)
    def raw_to_sample(self, raw_samples: torch.Tensor) -> torch.Tensor:
        return self.raw_to_mel_spec(raw_samples)

    def sample_to_raw(self, samples: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("mel spectrogram -> audio of the dual format goes through the MCLT diffusion decoder (outside this path)")

    # ---- un-mel for the diffusion decoder's conditioning (reference :259-271)
    def _unmel_weights(self):
        """Minimum-norm un-mel operator as a constant 1x1 conv (the reference solves lstsq(filters^T, mel) per call,
        frequency_scale.py:130-142): pseudo-inverse in float64, psd scale folded in."""
        if self._unmel is None:
            from ... import ops
            c = self.config
            if c.mdct_psd_num_bins != c.ms_num_stft_bins - 1:
                raise NotImplementedError("mdct_psd_num_bins != ms_num_stft_bins - 1 needs the second filter bank (ms_mdct_dual.py:148-160)")
            pinv = torch.linalg.pinv(self.ms_freq_scale.filters.double().t())          # (n_stft, n_mel)
            nb = c.mdct_psd_num_bins                                                     # (the last STFT bin is cropped, :264-265)
            w = torch.zeros((nb + 3) // 4 * 4, c.ms_num_frequencies, 1, 1)
            w[:nb, :, 0, 0] = (pinv[:nb] * c.mel_spec_to_mdct_psd_scale).float()
            wd = w.to(self.device).contiguous()
            self._unmel = ops.wprep(wd, 1, torch.float32, gain=math.sqrt(c.ms_num_frequencies), npix=0)
            self._unmel_keep = wd
        return self._unmel

    @torch.no_grad()
    def mel_spec_to_mdct_psd(self, mel_spec: torch.Tensor) -> torch.Tensor:
        """(B, C, 256, T) mel spectrogram -> (B, C, 2048, T) estimate of the MDCT power spectral density."""
        from ... import ops
        if self.device.type != "cuda":
            raise DDXError("MS_MDCT_DualFormat is not on a ROCm device: no CPU fallback")
        c = self.config
        x = mel_spec.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cn, n_mel, T = x.shape
        amp = torch.empty(B * Cn, 1, T, n_mel, device=self.device, dtype=torch.float32)
        # ((mel - offset).clip(0)) ** (1 / exponent), transposed to channel-last rows
        check(lib().ddx_mel_to_amplitude(ptr(x), ptr(amp), B * Cn, n_mel, T, 1.0, -float(c.raw_to_mel_spec_offset), 1.0 / float(c.ms_abs_exponent),
                                         current_stream()), "mel_to_amplitude")
        psd = ops.conv2d(amp, self._unmel_weights())                 # [B*C][1][T][nb]
        out = ops.nhwc_to_nchw(psd, channels=c.mdct_psd_num_bins)     # [B*C][nb][1][T]
        out = out.view(B, Cn, c.mdct_psd_num_bins, T)
        return out + c.mel_spec_to_mdct_psd_offset if c.mel_spec_to_mdct_psd_offset else out

    def raw_to_mdct(self, *a, **k):
        raise NotImplementedError("MDCT side of the dual format (utils/mclt.py) is outside the hot path built here")

    mdct_to_raw = raw_to_mdct
