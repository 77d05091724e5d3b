"""CPU: the oracle (oracle/edm2_oracle.py) against the golden vectors produced from the reference
(tools/make_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP path with it."""
import torch

from oracle import edm2_oracle as O
from tests.util import load_golden, rel_l2

TOL = 2e-6


def test_ops_golden():
    t, m = load_golden("ops")
    assert rel_l2(O.rms_normalize(t["normalize.w4.in"]), t["normalize.w4.out"]) < TOL
    assert rel_l2(O.rms_normalize(t["normalize.act.in"], [1]), t["normalize.act.out"]) < TOL
    assert rel_l2(O.rms_normalize(t["normalize.qk.in"], [2]), t["normalize.qk.out"]) < TOL
    act = t["normalize.act.in"]
    assert rel_l2(O.silu_mp(act), t["mp_silu.out"]) < TOL
    assert rel_l2(O.sum_mp(act, t["mp_sum.b"], 0.3), t["mp_sum.t03.out"]) < TOL
    assert rel_l2(O.sum_mp(act, t["mp_sum.b"], t["mp_sum.tt"]), t["mp_sum.tt.out"]) < TOL
    assert rel_l2(O.cat_mp(act, t["mp_cat.b"], 0.5), t["mp_cat.out"]) < TOL
    assert rel_l2(O.resample2x(t["resample.in"], "down"), t["resample.down.out"]) < TOL
    assert torch.equal(O.resample2x(t["resample.in"], "up"), t["resample.up.out"])
    for ch in (32, 128, 256):
        fr, ph = O.fourier_tables(ch)
        assert torch.equal(fr, t[f"mpfourier{ch}.freqs"]) and torch.equal(ph, t[f"mpfourier{ch}.phases"])
        assert rel_l2(O.fourier_mp(t[f"mpfourier{ch}.in"], fr, ph), t[f"mpfourier{ch}.out"]) < TOL
    for nm in ("c3g", "c1", "lin", "c3"):
        g = m[f"mpconv.{nm}.groups"]
        w, x, gain = t[f"mpconv.{nm}.w"], t[f"mpconv.{nm}.x"], t[f"mpconv.{nm}.gain"]
        for mode in ("eval", "train"):
            tr = mode == "train"
            assert rel_l2(O.conv_mp(x, w, groups=g, training=tr), t[f"mpconv.{nm}.{mode}.out"]) < 5e-6
            assert rel_l2(O.conv_mp(x, w, gain=gain, groups=g, training=tr), t[f"mpconv.{nm}.{mode}.out_gain"]) < 5e-6


def test_blocks_golden():
    t, m = load_golden("blocks")
    for nm, c in m["cases"].items():
        sd = {k[len(nm) + 1:]: v for k, v in t.items() if k.startswith(nm + ".blk.")}
        for mode in ("eval", "train"):
            y = O.block_forward(sd, "blk", t[f"{nm}.x"], t[f"{nm}.emb"], flavor=c["flavor"], resample=c["resample"],
                                attention=c["attn"], heads=c["cout"] // m["channels_per_head"], groups=8, training=mode == "train")
            assert rel_l2(y, t[f"{nm}.{mode}.out"]) < 5e-6, (nm, mode)


def _unet_case(name):
    t, m = load_golden(name)
    cfg = O.unet_cfg(**m["cfg"])
    if m["weights"] == "stored":
        sd = {k[3:]: v for k, v in t.items() if k.startswith("sd.")}
    else:
        sd = O.random_unet_state(cfg, m["seed"])
    return t, m, cfg, sd


def test_unet_tiny_golden():
    """BASELINE.json config 1: tiny EDM2 UNet, CPU float32."""
    t, m, cfg, sd = _unet_case("unet_tiny")
    mask = t["mask"].bool()
    emb = O.unet_embeddings(sd, cfg, t["clap"], mask)
    assert rel_l2(emb, t["embeddings"]) < TOL
    assert rel_l2(O.unet_sigma_logvar(sd, cfg, t["sigma"]), t["logvar"]) < TOL
    coll = {}
    out = O.unet_forward(sd, cfg, t["x_in"], t["sigma"], t["embeddings"], collect=coll)
    assert rel_l2(out, t["out"]) < 1e-5
    for k in [k for k in t if k.startswith("stage.")]:
        assert rel_l2(coll[k[6:]], t[k]) < 1e-5, k
    out2 = O.unet_forward(sd, cfg, t["x_in"], t["sigma"], t["embeddings"], x_ref=t["x_ref"], perturbed_input=t["perturbed_input"])
    assert rel_l2(out2, t["out_xref"]) < 1e-5
    sd_t = {k: (v * t[f"trainscale.{k}"].view(-1, *([1] * (v.ndim - 1))) if v.ndim >= 2 else v) for k, v in sd.items()}
    out_t = O.unet_forward(sd_t, cfg, t["x_in"], t["sigma"], t["embeddings"], training=True)
    assert rel_l2(out_t, t["out_train_unnormalized"]) < 1e-5
    assert O.unet_latent_shape(cfg, (2, 4, 37, 70)) == (2, 4, 36, 70)


def test_unet_seeded_golden():
    for name in ("unet_small", "unet_wide"):
        t, m, cfg, sd = _unet_case(name)
        out = O.unet_forward(sd, cfg, t["x_in"], t["sigma"], t["embeddings"])
        assert rel_l2(out, t["out"]) < 1e-5, name


def test_schedule_golden():
    t, _ = load_golden("schedule")
    for k, ref in t.items():
        parts = k[len("edm2."):].split(".")
        # key = edm2.<n>.<smax>.<smin>.<rho> with float fields containing one dot each
        n = int(parts[0])
        smax = float(parts[1] + "." + parts[2])
        smin = float(parts[3] + "." + parts[4])
        rho = float(parts[5] + "." + parts[6])
        assert rel_l2(O.schedule_edm2(n, smax, smin, rho), ref) < 1e-6


def test_sampler_golden():
    """diffusion_decode restatement (CFG + Heun + input perturbation) against the reference's output, injected noises."""
    t, m = load_golden("sampler")
    cfg = O.unet_cfg(**m["cfg"])
    sd = O.random_unet_state(cfg, m["seed"])
    emb = t["embeddings"]
    den = lambda x, s: O.unet_forward(sd, cfg, x, s, emb)
    for case, kw in m["cases"].items():
        noises = [t[f"{case}.noise{i}"] for i in range(m["num_steps"])]
        out, sig = O.sampler_edm2(den, tuple(m["shape"]), noises, num_steps=m["num_steps"], sigma_max=m["sigma_max"],
                                  sigma_min=m["sigma_min"], batch_size=m["B"], **kw)
        assert rel_l2(out, t[f"{case}.out"]) < 2e-5, case


def test_host_schedules_match_oracle():
    from dualdiffusion_amd.sampling.schedule import SamplingSchedule
    t, _ = load_golden("schedule")
    ref = t["edm2.100.200.0.0.03.7.0"]
    got = SamplingSchedule.get_schedule("edm2", 100, sigma_max=200.0, sigma_min=0.03, rho=7.0)
    assert torch.equal(got, ref)
    assert set(SamplingSchedule.get_schedules_list()) == {"edm2", "ln_linear", "linear", "cos", "scale_invariant"}


def test_vae_golden():
    t, m = load_golden("vae_small")
    cfg = O.vae_cfg(**m["cfg"])
    sd = O.random_vae_state(cfg, m["seed"])
    assert rel_l2(O.vae_embeddings(sd, t["labels_like"]), t["emb"]) < TOL
    mean, logvar = O.vae_encode(sd, cfg, t["x"], t["emb"], tuple(m["freq_range"]))
    assert rel_l2(mean, t["latents"]) < 1e-5 and abs(logvar - float(t["noise_logvar"])) < 1e-6
    assert rel_l2(O.vae_decode(sd, cfg, t["latents"], t["emb"], tuple(m["freq_range"])), t["recon"]) < 1e-5


def test_mel_golden_and_band_edges():
    """mel-STFT oracle against the reference output; the integer band-edge table is bit-exact."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    t, m = load_golden("mel_stft")
    fb = M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000)
    mel = M.raw_to_mel(t["audio"], window=M.hann_power_window(6400, 32.0), hop=256, filters=fb)
    assert rel_l2(mel, t["mel"]) < 2e-5
    assert int((fb > 0).sum()) == m["nnz"] and torch.equal(fb.sum(dim=0), t["filter_colsum"])
    host = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)
    assert torch.equal(host.band_edges(), t["band_edges"])          # product-side table == reference non-zero support
    assert torch.equal(host.filters, fb)


def test_mss_loss_golden():
    """MSS loss oracle (value and gradient) against fixtures produced by the reference's MSSLoss2D."""
    from oracle import mss_oracle as M
    t, m = load_golden("mss_loss")
    for name, kw in m.items():
        if not isinstance(kw, dict):
            continue                      # provenance entries (torch version, thread count)
        kw = dict(kw)
        kw["block_widths"] = tuple(kw["block_widths"])
        loss, grad = M.mss_loss_and_grad(t[f"{name}.sample"], t[f"{name}.target"], **kw)
        assert rel_l2(loss, t[f"{name}.loss"]) < 1e-6, name
        assert rel_l2(grad, t[f"{name}.grad"]) < 1e-5, name


def test_unet_train_golden():
    """Train-batch oracle (loss + parameter gradients by autograd) against the fixture produced by the reference UNet in train
    mode with the loss lines of UNetTrainer.unet_train_batch."""
    t, m = load_golden("unet_train")
    cfg = O.unet_cfg(**m["cfg"])
    sd = O.random_unet_state(cfg, seed=m["seed"], gain_value=m["gain_value"], normalized=False)
    params = {k: sd[k].clone().requires_grad_(True) for k in m["grads"]}
    sd_o = dict(sd); sd_o.update(params)
    loss = O.unet_train_loss(sd_o, cfg, t["samples"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), t["pert"], m["input_perturbation"],
                             tuple(m["freq_range"]))
    assert rel_l2(loss, t["loss"]) < 1e-5
    grads = torch.autograd.grad(loss.mean(), list(params.values()))
    for k, g in zip(params, grads):
        assert rel_l2(g, t[f"grad.{k}"]) < 1e-3, k


def test_ddec_golden():
    """Diffusion-decoder oracle against the reference's forward (bit-equal in the reference's bfloat16 op sequence)."""
    from oracle import ddec_oracle as DO
    t, m = load_golden("ddec_small")
    cfg = DO.ddec_cfg(**m["cfg"])
    sd = DO.random_ddec_state(cfg, m["seed"])
    coll = {}
    out = DO.ddec_forward(sd, cfg, t["x_in"], t["sigma"], t["x_ref"], collect=coll)
    assert rel_l2(out, t["out"]) < 1e-6
    for k in [k for k in t if k.startswith("stage.")]:
        assert rel_l2(coll[k[6:]].float(), t[k]) < 1e-6, k
    assert rel_l2(DO.ddec_forward(sd, cfg, t["x_in"], t["sigma"], t["x_ref"], compute_dtype=torch.float32), t["out_fp32_oracle"]) < 1e-6


def test_ema_step_golden():
    """Parameter pass after the backward (clip + AdamW + three EMAs incl. power-function and feedback + forced weight norm): the
    restatement against the reference's torch.optim.AdamW + EMA_Manager.update + normalize run (tests/golden/ema_step)."""
    from oracle import train_oracle as TO
    t, m = load_golden("ema_step")
    names = [k[3:] for k in t if k.startswith("p0.")]
    p = {k: t[f"p0.{k}"].clone() for k in names}
    mm, vv = {k: torch.zeros_like(x) for k, x in p.items()}, {k: torch.zeros_like(x) for k, x in p.items()}
    emas = [({k: x.clone() for k, x in p.items()}, c.get("feedback_beta")) for c in m["emas"].values()]
    for s in range(m["steps"]):
        grads = {k: t[f"g{s}.{k}"] for k in names}
        norm = TO.adamw_ema_wn_step(p, grads, mm, vv, s + 1, m["lr"], m["loss_scale"], m["max_norm"], emas, m["betas"][s], set(m["wn"]))
        assert abs(norm - m["norms"][s]) / m["norms"][s] < 1e-5
    for k in names:
        assert rel_l2(p[k], t[f"p{m['steps']}.{k}"]) < 2e-6, k
    for (name, _c), (et, _fb) in zip(m["emas"].items(), emas):
        for k in names:
            assert rel_l2(et[k], t[f"ema_{name}.{k}"]) < 2e-6, (name, k)
    assert abs(TO.power_function_beta(0.05, 32, 16) - m["betas"][1][1]) < 1e-12


def test_ms_mel_spec_golden():
    """MS_MDCT_DualFormat.raw_to_mel_spec (dual-window mel spectrogram of the live format) and mel_spec_to_mdct_psd: the restatement
    against the reference's outputs; the slaney bank's integer band edges through the host table class."""
    from oracle import mel_oracle as M
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    t, m = load_golden("ms_mel_spec")
    assert rel_l2(M.raw_to_ms_mel_spec(t["audio"]), t["mel"]) < 2e-5
    assert rel_l2(M.ms_mel_to_mdct_psd(t["mel"])[..., ::16], t["mdct_psd_frames16"]) < 2e-3
    fs = FrequencyScale("mel", 0.0, 16000.0, 32000, 2049, 256, "slaney")
    assert torch.equal(fs.band_edges(), t["band_edges"]) and int((fs.filters > 0).sum()) == m["nnz"] == 4067
    assert rel_l2(fs.filters.sum(dim=0), t["filter_colsum"]) < 1e-7
    assert torch.equal(fs.filters, M.slaney_mel_filterbank(2049, 256, 0.0, 16000.0, 32000))


def test_dae_g1_golden():
    """DAE_G1 (the live autoencoder: (1,3,3)/(2,3,3)/(1,5,5) stereo-depth kernels, axis-folded attention, tiled encode): the restatement
    against the reference module's outputs (tests/golden/dae_g1_small)."""
    from oracle import dae_oracle as DO
    t, m = load_golden("dae_g1_small")
    cfg = DO.dae_cfg(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in m["cfg"].items()})
    sd = DO.random_dae_state(cfg, m["seed"])
    emb = DO.dae_embeddings(sd, t["emb_in"])
    assert rel_l2(emb, t["emb"]) < 2e-6
    assert rel_l2(DO.dae_encode(sd, cfg, t["x"], emb), t["latents"]) < 1e-5
    assert rel_l2(DO.dae_encode(sd, cfg, t["x"], emb, normalize_latents=False), t["latents_raw"]) < 1e-5
    assert rel_l2(DO.dae_decode(sd, cfg, t["latents"], emb), t["recon"]) < 1e-5
    assert rel_l2(DO.dae_tiled_encode(sd, cfg, t["x_tiled"], emb[:1], **m["tiled"]), t["latents_tiled"]) < 1e-5


def test_train_options_golden():
    """Oracle train batch with the trainer / model options that are off by default -- dropout (recorded keep masks of the reference's draws),
    conditioning_perturbation, normalize_latents, use_dynamic_sigma_data, x_ref under autograd -- against the reference's loss, parameter
    gradients and d loss / d x_ref (tests/golden/unet_train_options.safetensors; unet_trainer.py:203-296, unet_edm2_b4.py:124-125,293-294)."""
    t, m = load_golden("unet_train_options")
    cfg = O.unet_cfg(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in m["cfg"].items()})
    sd = O.random_unet_state(cfg, seed=m["seed"], gain_value=m["gain_value"], normalized=False)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "fourier" not in k}
    sd_o = dict(sd); sd_o.update(params)
    xr = t["x_ref"].clone().requires_grad_(True)
    masks = {k[len("dropout_mask."):]: v.bool() for k, v in t.items() if k.startswith("dropout_mask.")}
    assert abs(torch.cat([v.flatten().float() for v in masks.values()]).mean().item() - (1 - cfg["dropout"])) < 0.02
    loss = O.unet_train_loss(sd_o, cfg, t["latents"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), t["pert"], m["input_perturbation"],
                             conditioning_perturbation=t["cpert"], conditioning_perturbation_scale=m["conditioning_perturbation"],
                             normalize_latents=m["normalize_latents"], dynamic_sigma_data=tuple(m["dynamic_sigma_data"]), ref_samples=xr,
                             dropout_masks=masks)
    assert rel_l2(loss.detach(), t["loss"]) < 1e-5
    names = m["grads"]
    g = torch.autograd.grad(loss.mean(), [params[k] for k in names] + [xr])
    for k, gk in zip(names, g[:-1]):
        assert rel_l2(gk, t[f"grad.{k}"]) < 1e-3, k
    assert rel_l2(g[-1], t["grad_x_ref"]) < 1e-4
    # every option changes the objective: switching one off moves the loss
    base = float(loss.detach().mean())
    off = O.unet_train_loss(sd, cfg, t["latents"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), t["pert"], m["input_perturbation"],
                            normalize_latents=True, dynamic_sigma_data=tuple(m["dynamic_sigma_data"]), ref_samples=t["x_ref"], dropout_masks=masks)
    assert abs(float(off.mean()) - base) > 1e-6 * abs(base)
