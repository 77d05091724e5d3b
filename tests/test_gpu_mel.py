"""GPU parity of the fused mel-STFT kernel (FFT-6400 in LDS + banded mel) against the reference's SpectrogramFormat output."""
import pytest
import torch

from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _fmt():
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    return SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")


def test_raw_to_sample_matches_reference():
    t, m = load_golden("mel_stft")
    fmt = _fmt()
    mel = fmt.raw_to_sample(t["audio"])
    assert mel.shape == t["mel"].shape and mel.dtype == torch.float32
    e = rel_l2(mel, t["mel"])
    print(f"mel-STFT rel-L2 vs reference: {e:.3e}")
    assert e < 1e-4, e
    # mono input (imaginary lane empty) and ragged frame counts (T not a multiple of the 8 frames per workgroup)
    a = t["audio"][:1, :1, :20000 + 256 * 3]
    from oracle import mel_oracle as M
    ref = M.raw_to_mel(a, window=M.hann_power_window(6400, 32.0), hop=256, filters=M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000))
    got = fmt.raw_to_sample(a)
    assert got.shape == ref.shape and rel_l2(got, ref) < 1e-4


def test_mel_linearity_and_silence():
    """Size-independent properties at a larger size: |STFT| mel is positively homogeneous before the exponent, and
    digital silence maps to the constant (0 - mean) * scale."""
    fmt = _fmt()
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1, 2, 256 * 500, generator=g) * 0.05
    c = fmt.config
    m1 = fmt.raw_to_sample(a) / c.raw_to_sample_scale + c.sample_mean
    m2 = fmt.raw_to_sample(a * 16.0) / c.raw_to_sample_scale + c.sample_mean
    assert rel_l2(m2, m1 * 2.0) < 1e-4                     # 16 ** 0.25 = 2
    z = fmt.raw_to_sample(torch.zeros(1, 2, 256 * 40))
    assert torch.allclose(z.cpu(), torch.full_like(z.cpu(), -c.sample_mean * c.raw_to_sample_scale), atol=1e-6)
