"""BASELINE.json configs[2] as a parity case: audio -> mel-STFT -> VAE.encode(mode) -> UNet train batch (forward, EDM2 loss,
backward) + multi-scale spectral loss on VAE.decode(latents) vs the mel spectrogram -- every stage on the HIP kernels, compared
with the same chain through the CPU oracles (each of which is pinned to the reference by its own golden fixture)."""
import pytest
import torch

from oracle import edm2_oracle as O
from oracle import mel_oracle as M
from oracle import mss_oracle as MS
from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


class _Fmt:
    def __init__(self):
        from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
        self.ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)
        self.fs = self.ms_freq_scale

    def get_ln_freqs(self, x):
        ln = self.fs.get_unscaled(x.shape[2] + 2, device=x.device)[1:-1].log2()
        ln = ln.view(1, 1, -1, 1).repeat(x.shape[0], 1, 1, x.shape[3])
        return ((ln - ln.mean()) / ln.std()).to(x.dtype)


def test_config3_mel_vae_unet_train_mss():
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.modules.vaes.vae_edm2 import AutoencoderKL_EDM2, DualDiffusionVAE_EDM2Config
    from dualdiffusion_amd.training.loss.multiscale_spectral import MSSLoss2D, MSSLoss2DConfig
    from dualdiffusion_amd.training.unet_grad import UNetTrainer
    g = torch.Generator().manual_seed(77)
    B, T = 2, 32
    audio = torch.randn(B, 2, 256 * (T - 1), generator=g) * 0.1
    # ---- oracle chain (CPU fp32)
    fcfg = SpectrogramFormatConfig()
    mel_ref = M.raw_to_mel(audio, window=M.hann_power_window(6400, 32.0), hop=256, filters=M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000),
                           exponent=fcfg.abs_exponent, mean=fcfg.sample_mean, scale=fcfg.raw_to_sample_scale)
    _t, vm = load_golden("vae_small")
    vcfg = O.vae_cfg(**vm["cfg"])
    vsd = O.random_vae_state(vcfg, vm["seed"])
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    ucfg = O.unet_cfg(**over)
    usd = O.random_unet_state(ucfg, seed=5, gain_value=0.5, normalized=False)
    labels = torch.randn(B, vm["cfg"]["label_dim"], generator=g)
    clap = torch.randn(B, 64, generator=g)
    sigma = torch.tensor([0.6, 3.0])
    mask = torch.tensor([True, False])
    # ---- HIP chain
    fmt = SpectrogramFormat(fcfg).to(device="cuda")
    mel = fmt.raw_to_sample(audio)                                                    # [B, 2, 256, T]
    assert mel.shape == (B, 2, 256, T) and rel_l2(mel, mel_ref) < 1e-4
    vae = AutoencoderKL_EDM2(DualDiffusionVAE_EDM2Config(**vm["cfg"])).requires_grad_(False).train(False)
    vae.load_state_dict(vsd)
    vae = vae.to(device="cuda", dtype=torch.float32)
    ffmt = _Fmt()
    with torch.no_grad():
        vemb = vae.get_embeddings(labels, labels_like=labels)
        latents = vae.encode(mel, vemb, ffmt).mode()
        recon = vae.decode(latents, vemb, ffmt)
    vemb_ref = O.vae_embeddings(vsd, labels)
    lat_ref, _ = O.vae_encode(vsd, vcfg, mel_ref, vemb_ref)
    rec_ref = O.vae_decode(vsd, vcfg, lat_ref, vemb_ref)
    assert rel_l2(latents, lat_ref) < 1e-4 and rel_l2(recon, rec_ref) < 1e-4
    noise = torch.randn(lat_ref.shape, generator=g)
    unet = UNet(UNetConfig(**over)).requires_grad_(False)
    unet.load_state_dict(usd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    loss, grads = UNetTrainer(unet).train_batch(latents, clap, sigma, noise, mask, ffmt)
    params = {k: v.clone().requires_grad_(True) for k, v in usd.items() if v.is_floating_point() and "fourier" not in k}
    sd_o = dict(usd); sd_o.update(params)
    loss_ref = O.unet_train_loss(sd_o, ucfg, lat_ref, clap, sigma, noise, mask)
    gref = dict(zip(params, torch.autograd.grad(loss_ref.mean(), list(params.values()))))
    assert rel_l2(loss, loss_ref) < 1e-2
    worst = max(rel_l2(grads[k].reshape(gref[k].shape), gref[k]) for k in gref if "gain" not in k)
    mss = MSSLoss2D(MSSLoss2DConfig(), torch.device("cuda"))
    recon_g = recon.detach().clone().requires_grad_(True)
    ml = mss.mss_loss(recon_g, mel)
    ml.sum().backward()
    ml_ref, mg_ref = MS.mss_loss_and_grad(rec_ref, mel_ref)
    print(f"config 3 chain: latents {rel_l2(latents, lat_ref):.1e}, recon {rel_l2(recon, rec_ref):.1e}, unet loss {rel_l2(loss, loss_ref):.1e}, "
          f"worst unet gradient {worst:.1e}, mss loss {rel_l2(ml, ml_ref):.1e}, mss grad {rel_l2(recon_g.grad, mg_ref):.1e}")
    assert worst < 3e-2
    assert rel_l2(ml, ml_ref) < 1e-4 and rel_l2(recon_g.grad, mg_ref) < 1e-3
