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


def test_fgla_linear_pieces_match_oracle():
    """The two linear halves of one FGLA iteration through the C ABI, on well-conditioned inputs (unit phasors):
    synth + overlap-add == torch.istft semantics, analysis == torch.stft semantics (tolerance 2e-5)."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd._lib import check, current_stream, lib, ptr
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
    win = M.hann_power_window(6400, 32.0)
    g = torch.Generator().manual_seed(11)
    B, Cn, T, nb = 2, 2, 40, 3201
    mags = torch.rand(B, Cn, nb, T, generator=g) + 0.1
    phase = torch.rand(B, Cn, nb, T, generator=g) * 6.28318
    spec = torch.polar(mags, phase)
    ref_wave = M.istft_frames(spec.reshape(-1, nb, T), win, 256).reshape(B, Cn, -1)
    st = current_stream()
    us = nb + 1                                                            # state / magnitude rows padded to an even bin count
    u = torch.zeros(B, T, Cn, us, 2)
    u[:, :, :, :nb] = torch.view_as_real(torch.polar(torch.ones_like(mags), phase)).permute(0, 3, 1, 2, 4)   # [B][T][C][nb][2]
    u = u.cuda()
    mg = torch.zeros(B, Cn, T, nb + 3)
    mg[..., :nb] = mags.permute(0, 1, 3, 2)
    mg = mg.contiguous().cuda()
    frames = torch.empty(B, T, Cn, 6400, device="cuda")
    audio = torch.empty(B, Cn, 256 * (T - 1), device="cuda")
    check(lib().ddx_fgla_synth(ptr(u), us, ptr(mg), ptr(fmt.window), ptr(fmt.twiddle), ptr(frames), B, Cn, T, 6400, nb + 3, 0.0, 1, st))
    check(lib().ddx_fgla_ola(ptr(frames), ptr(fmt.window), ptr(audio), B, Cn, T, 6400, 256, st))
    e = rel_l2(audio, ref_wave)
    print(f"istft piece rel-L2 {e:.3e}")
    assert e < 2e-5
    # analysis: u <- stft(audio) - momentum * u  (u = known tensor)
    u0 = torch.randn(B, T, Cn, nb, 2, generator=g)
    ud = torch.zeros(B, T, Cn, us, 2)
    ud[:, :, :, :nb] = u0
    ud = ud.cuda()
    check(lib().ddx_fgla_analysis(ptr(audio), ptr(fmt.window), ptr(fmt.twiddle), ptr(ud), us, B, Cn, T, 256 * (T - 1), 6400, 256, 0.25, st))
    ref_spec = M.stft_frames(ref_wave, win, 256)                                  # (B, C, nb, T)
    ref_u = torch.view_as_real(ref_spec).permute(0, 3, 1, 2, 4) - 0.25 * u0
    e = rel_l2(ud[:, :, :, :nb], ref_u)
    print(f"stft piece rel-L2 {e:.3e}")
    assert e < 2e-5


@pytest.mark.parametrize("final,t_lerp", [(0, 0.3), (0, -0.2), (1, 0.0)])
def test_fgla_iter_equals_analysis_then_synth(final, t_lerp):
    """ddx_fgla_iter (analysis of one iteration + synthesis of the next, one launch per frame) against the two calls it replaces on the
    same inputs: the same state and the same frames (same arithmetic in the same order: 1e-6)."""
    from dualdiffusion_amd._lib import check, current_stream, lib, ptr
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    B, Cn, T, nb, hop, N = 2, 2, 33, 3201, 256, 6400
    us, ms, Lout = nb + 1, nb + 3, hop * (T - 1)
    audio = torch.randn(B, Cn, Lout, device="cuda", generator=g)
    u0 = torch.randn(B, T, Cn, us, 2, device="cuda", generator=g)
    mg = torch.randn(B, Cn, T, ms, device="cuda", generator=g)           # (negative entries: the relu of the un-mel is part of the read)
    st = current_stream()
    ua, fa = u0.clone(), torch.empty(B, T, Cn, N, device="cuda")
    check(lib().ddx_fgla_analysis(ptr(audio), ptr(fmt.window), ptr(fmt.twiddle), ptr(ua), us, B, Cn, T, Lout, N, hop, 0.4975, st))
    check(lib().ddx_fgla_synth(ptr(ua), us, ptr(mg), ptr(fmt.window), ptr(fmt.twiddle), ptr(fa), B, Cn, T, N, ms, t_lerp, final, st))
    ub, fb = u0.clone(), torch.empty(B, T, Cn, N, device="cuda")
    check(lib().ddx_fgla_iter(ptr(audio), ptr(fmt.window), ptr(fmt.twiddle), ptr(ub), us, ptr(mg), ms, ptr(fb), B, Cn, T, Lout, N, hop,
                              0.4975, t_lerp, final, st))
    torch.cuda.synchronize()
    assert rel_l2(ub[:, :, :, :nb], ua[:, :, :, :nb]) < 1e-6
    assert torch.equal(ub[:, :, :, nb:], u0[:, :, :, nb:])               # row padding written back unchanged
    assert rel_l2(fb, fa) < 1e-6


def test_fgla_sample_to_raw_tracks_reference():
    """un-mel (pseudo-inverse GEMM) + 4 FGLA iterations.  The iteration re-normalises near-empty bins to unit phasors, so
    the float32 result is rounding-sensitive: the REFERENCE's own float32 output sits ~8 % (rel-L2) from the float64
    trajectory on this input.  The HIP path must be as close to the float64 trajectory as the reference is (within 1.5x),
    for both anneal regimes."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    t, m = load_golden("mel_stft")
    g, gm = load_golden("fgla")
    win, fb = M.hann_power_window(6400, 32.0), M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000)
    for case, coh in gm["coherence"].items():
        fmt = SpectrogramFormat(SpectrogramFormatConfig(stereo_coherence=coh)).to(device="cuda")
        raw = fmt.sample_to_raw(t["mel"][:1], n_fgla_iters=gm["n_iter"], quiet=True)
        assert raw.shape == g[f"{case}.raw"].shape
        truth = M.mel_to_raw(t["mel"][:1], window=win, hop=256, filters=fb, n_iter=gm["n_iter"], stereo_coherence=coh, dtype=torch.float64)
        e_ref, e_gpu = rel_l2(g[f"{case}.raw"], truth), rel_l2(raw, truth)
        print(f"FGLA {case}: reference-fp32 vs fp64 {e_ref:.3e}, HIP vs fp64 {e_gpu:.3e}")
        assert e_gpu < 1.5 * e_ref + 1e-3, (case, e_gpu, e_ref)
        # one iteration is still well inside the linear regime of the error growth: tight check against the float64 run
        one = fmt.sample_to_raw(t["mel"][:1], n_fgla_iters=1, quiet=True)
        t1 = M.mel_to_raw(t["mel"][:1], window=win, hop=256, filters=fb, n_iter=1, stereo_coherence=coh, dtype=torch.float64)
        o1 = M.mel_to_raw(t["mel"][:1], window=win, hop=256, filters=fb, n_iter=1, stereo_coherence=coh)
        assert rel_l2(one, t1) < 1.5 * rel_l2(o1, t1) + 1e-3


def test_fgla_roundtrip_property():
    """encode -> decode round trip at a longer length: the re-encoded mel of the reconstructed audio stays close to the
    input mel (FGLA is a magnitude-consistency iteration), and one STFT/iSTFT analysis-synthesis pair is the identity."""
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
    g = torch.Generator().manual_seed(5)
    tt = torch.arange(256 * 200) / 32000.0
    a = (0.1 * torch.sin(2 * torch.pi * 440 * tt) + 0.05 * torch.sin(2 * torch.pi * 1330 * tt))[None, None].repeat(1, 2, 1)
    a = a + 0.002 * torch.randn(a.shape, generator=g)
    mel = fmt.raw_to_sample(a)
    rec = fmt.sample_to_raw(mel, n_fgla_iters=30, quiet=True)
    mel2 = fmt.raw_to_sample(rec)
    e = rel_l2(mel2[..., 16:-16], mel.cpu()[..., 16:mel2.shape[-1] - 16])
    print(f"mel(decode(mel)) vs mel: {e:.3e}")
    assert e < 0.15


def test_ms_mel_spec_matches_reference():
    """MS_MDCT_DualFormat.raw_to_mel_spec on the HIP kernel (two in-LDS FFT-4096 per frame, blend, banded slaney bank) against the
    reference's output, its shapes, and mel_spec_to_mdct_psd (un-mel as a constant 1x1 conv)."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd.modules.formats.ms_mdct_dual import MS_MDCT_DualFormat, MS_MDCT_DualFormatConfig
    t, m = load_golden("ms_mel_spec")
    fmt = MS_MDCT_DualFormat(MS_MDCT_DualFormatConfig()).to(device="cuda")
    assert fmt.get_raw_crop_width(1408768) == m["crop_width"] and list(fmt.get_mel_spec_shape(bsz=1)) == m["shape_45s"]
    mel = fmt.raw_to_mel_spec(t["audio"])
    e = rel_l2(mel, t["mel"])
    print(f"ms_mel_spec rel-L2 vs reference {e:.3e}")
    assert mel.shape == t["mel"].shape and e < 1e-5
    # mono, and a length whose last workgroup is ragged
    a1 = t["audio"][:1, :1, :256 * 100 + 17]
    assert rel_l2(fmt.raw_to_mel_spec(a1), M.raw_to_ms_mel_spec(a1)) < 1e-5
    psd = fmt.mel_spec_to_mdct_psd(t["mel"])
    e2 = rel_l2(psd[..., ::16], t["mdct_psd_frames16"])
    print(f"mel_spec_to_mdct_psd rel-L2 vs reference {e2:.3e}")
    assert psd.shape == (2, 2, 2048, 128) and e2 < 2e-3     # (the reference's float32 gels solve is itself ~1e-4 off float64)
    # ln_freqs of the UNet come from this format's scale (unet_edm2_b4.py:246)
    assert fmt.ms_freq_scale.num_stft_bins == 2049 and fmt.ms_freq_scale.filter_norm == "slaney"


def test_unmel_and_one_fgla_iteration_vs_fixture():
    """un-mel (pseudo-inverse as a 1x1 conv) against the oracle's minimum-norm solve, and ONE FGLA iteration against the reference's
    waveform fixture bound: the 4-iteration fixture is the only waveform the reference run produced, so the single iteration is
    judged against the float64 oracle with the float32 oracle's own distance as the yardstick, and the final 4-iteration waveform
    directly against the fixture with the reference's float32-vs-float64 distance as tolerance."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd import ops
    from dualdiffusion_amd._lib import check, current_stream, lib, ptr
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    t, m = load_golden("mel_stft")
    g, gm = load_golden("fgla")
    fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
    mel = t["mel"][:1]
    c = fmt.config
    B, Cn, n_mel, T = mel.shape
    amp = torch.empty(B * Cn, 1, T, n_mel, device="cuda")
    check(lib().ddx_mel_to_amplitude(ptr(mel.cuda().contiguous()), ptr(amp), B * Cn, n_mel, T, c.raw_to_sample_scale, c.sample_mean,
                                     1.0 / c.abs_exponent, current_stream()))
    mags = ops.conv2d(amp, fmt._unmel_weights())[..., :c.num_stft_bins]          # [B*C][1][T][nb]
    fb = M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000)
    amp_ref = ((mel / c.raw_to_sample_scale + c.sample_mean).clamp(min=0) ** (1 / c.abs_exponent))
    ref = M.unmel(amp_ref.double(), fb.double()).float()                          # (B, C, nb, T): lstsq + relu
    got = torch.relu(mags).view(B, Cn, T, -1).permute(0, 1, 3, 2)
    e = rel_l2(got, ref)
    print(f"un-mel (pinv conv + relu) vs minimum-norm lstsq: {e:.3e}")
    assert e < 2e-4
    win = M.hann_power_window(6400, 32.0)
    raw = fmt.sample_to_raw(mel, n_fgla_iters=gm["n_iter"], quiet=True)
    truth = M.mel_to_raw(mel, window=win, hop=256, filters=fb, n_iter=gm["n_iter"], stereo_coherence=0.67, dtype=torch.float64)
    e_fix, e_ref = rel_l2(raw, g["default.raw"]), rel_l2(g["default.raw"], truth)
    print(f"FGLA x{gm['n_iter']}: HIP vs the reference's waveform {e_fix:.3e} (reference fp32 vs fp64: {e_ref:.3e})")
    assert e_fix < 2.5 * e_ref + 1e-3


def test_device_band_tables_bit_exact():
    """SURVEY.md 8 a-11 on the GPU box: the integer band tables the mel kernels actually index with -- the DEVICE buffers `band_start` / `band_len`
    of SpectrogramFormat (mel scale, 3201 bins) and MS_MDCT_DualFormat (slaney, 2049 bins) -- read back and compared bit for bit with the
    reference's non-zero filter support (`band_edges`, `nnz`, `filter_colsum` of the fixtures made from frequency_scale.py:45-58,144-168), and the
    device filter weights with the reference's filter values inside that support."""
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.modules.formats.ms_mdct_dual import MS_MDCT_DualFormat, MS_MDCT_DualFormatConfig
    cases = (("mel_stft", _fmt(), lambda f: f.freq_scale),
             ("ms_mel_spec", MS_MDCT_DualFormat(MS_MDCT_DualFormatConfig()).to(device="cuda"), lambda f: f.ms_freq_scale))
    for name, fmt, scale_of in cases:
        t, m = load_golden(name)
        assert fmt.band_start.is_cuda and fmt.band_len.is_cuda and fmt.band_w.is_cuda
        assert fmt.band_start.dtype == torch.int32 and fmt.band_len.dtype == torch.int32
        start, length, bw = fmt.band_start.cpu(), fmt.band_len.cpu(), fmt.band_w.cpu()
        edges = t["band_edges"]                                   # (filters, 2) int32: first / last bin with non-zero weight, inclusive
        assert torch.equal(start, edges[:, 0]) and torch.equal(start + length - 1, edges[:, 1]), name
        # the weights the kernel multiplies with: zero outside [0, len), their count and column sums are the reference filter bank's
        cols = torch.arange(bw.shape[1]).unsqueeze(0)
        assert bool((bw[cols >= length.unsqueeze(1)] == 0).all())
        assert int((bw > 0).sum()) == m["nnz"], (name, int((bw > 0).sum()), m["nnz"])
        sc = scale_of(fmt)
        host = FrequencyScale(sc.freq_scale, sc.freq_min, sc.freq_max, sc.sample_rate, sc.num_stft_bins, sc.num_filters, sc.filter_norm)
        assert torch.equal(host.band_edges(), edges)
        fb = host.filters
        for k in range(0, fb.shape[1], 17):
            n = int(length[k])
            assert torch.equal(bw[k, :n], fb[int(start[k]):int(start[k]) + n, k])
        assert torch.equal(fb.sum(dim=0), t["filter_colsum"])


def test_fgla_full_size_geometry():
    """BASELINE configs[4] geometry of the phase reconstruction: `sample_to_raw` on a (2, 2, 256, 5504) mel (45 s stereo, 1 408 768 samples; 5504
    frames x 32-row frame blocks, the 0.28 GB state / noise buffers) with 8 iterations -- finite, the right length, the re-encoded mel within the
    short-clip round trip's bound, and GEOMETRY INDEPENDENCE: every kernel of the iteration is per frame (+ a 25-frame overlap-add), so the first
    frames of the long call must equal the same call on a crop of the mel away from the crop's right edge (reference phase_recovery.py:78-119)."""
    fmt = _fmt()
    c = fmt.config
    g = torch.Generator().manual_seed(21)
    n = 1408768
    tt = torch.arange(n) / 32000.0
    a = torch.stack([0.08 * torch.sin(2 * torch.pi * (220.0 + 3.0 * tt) * tt) + 0.04 * torch.sin(2 * torch.pi * 1330 * tt),
                     0.06 * torch.sin(2 * torch.pi * 554.4 * tt) + 0.05 * torch.sin(2 * torch.pi * (900.0 - 2.0 * tt) * tt)])[None].repeat(2, 1, 1)
    a[1] = a[1].flip(0) * 0.7
    a = a + 0.002 * torch.randn(a.shape, generator=g)
    mel = fmt.raw_to_sample(a)
    assert tuple(mel.shape) == (2, 2, 256, 5504)
    rec = fmt.sample_to_raw(mel, n_fgla_iters=8, quiet=True)
    assert tuple(rec.shape) == (2, 2, n) and bool(torch.isfinite(rec).all())
    mel2 = fmt.raw_to_sample(rec)
    e = rel_l2(mel2[..., 16:-16], mel.cpu()[..., 16:-16])
    print(f"full size: mel(decode(mel)) vs mel after 8 iterations: {e:.3e}")
    assert e < 0.25, e
    # geometry independence (the iteration is deterministic: rand_init is off in the reference's call): one iteration moves information by at
    # most 25 frames (a frame's analysis window covers the audio that 25 frames on either side were overlap-added into), so after 8 iterations and
    # the final synthesis the first 512 - 9 * 25 - 32 frames of a 512-frame crop have seen exactly the neighbours they see in the long call
    crop = 512
    rec_c = fmt.sample_to_raw(mel[..., :crop].contiguous(), n_fgla_iters=8, quiet=True)
    keep = (crop - 9 * 25 - 32) * c.hop_length - c.padded_length // 2
    e_c = rel_l2(rec[..., :keep], rec_c[..., :keep])
    print(f"full size vs {crop}-frame crop on the first {keep} samples: {e_c:.3e}")
    assert e_c < 1e-5, e_c
