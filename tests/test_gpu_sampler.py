"""GPU parity of the sampler loop (HIP UNet + ddx_lincomb3 step algebra) against the reference's diffusion_decode output."""
import pytest
import torch

from oracle import edm2_oracle as O
from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def test_diffusion_decode_matches_reference():
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.pipelines.dual_diffusion_pipeline import DualDiffusionPipeline, SampleParams
    t, m = load_golden("sampler")
    cfg = O.unet_cfg(**m["cfg"])
    unet = UNet(UNetConfig(**m["cfg"])).requires_grad_(False).train(False)
    unet.load_state_dict(O.random_unet_state(cfg, m["seed"]))
    unet = unet.to(device="cuda", dtype=torch.float32)

    class Fmt:
        ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)
    pipe = DualDiffusionPipeline({"unet": unet, "format": Fmt()})
    for case, kw in m["cases"].items():
        params = SampleParams(seed=1234, num_steps=m["num_steps"], batch_size=m["B"], sigma_max=m["sigma_max"], sigma_min=m["sigma_min"],
                              sigma_data=1.0, rho=7.0, schedule="edm2", **kw)
        noises = [t[f"{case}.noise{i}"] for i in range(m["num_steps"])]
        out = pipe.diffusion_decode(params, quiet=True, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]), noises=noises)
        e = rel_l2(out, t[f"{case}.out"])
        print(f"sampler {case}: rel-L2 {e:.3e}")
        assert e < 1e-4, (case, e)
    assert rel_l2(torch.tensor(pipe.debug_info["sigma_schedule"]), O.schedule_edm2(m["num_steps"], m["sigma_max"], m["sigma_min"])) < 1e-6
    # generator path runs and is deterministic per seed
    params = SampleParams(seed=5, num_steps=2, batch_size=m["B"], sigma_max=20.0, sigma_min=0.1, sigma_data=1.0)
    a = pipe.diffusion_decode(params, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]))
    b = pipe.diffusion_decode(params, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]))
    assert torch.equal(a, b) and torch.isfinite(a).all()
    # the one-graph-per-step loop (default) and the eager step loop are the same arithmetic: bit-identical, with and without
    # ancestral noise, Heun on and off
    for kw in (dict(use_heun=True, input_perturbation=1.0), dict(use_heun=False, input_perturbation=0.0)):
        params = SampleParams(seed=7, num_steps=3, batch_size=m["B"], sigma_max=20.0, sigma_min=0.1, sigma_data=1.0, cfg_scale=1.7, **kw)
        assert pipe.step_graph
        g_out = pipe.diffusion_decode(params, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]))
        pipe.step_graph = False
        try:
            e_out = pipe.diffusion_decode(params, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]))
        finally:
            pipe.step_graph = True
        assert torch.equal(g_out, e_out), kw


def test_sampler_device_scalar_ops():
    """ddx_sampler_load / ddx_lincomb3_dev / ddx_step_advance against their host-scalar twins."""
    from dualdiffusion_amd import ops
    g = torch.Generator().manual_seed(3)
    B, nb, shp = 2, 4, (2, 3, 4, 6)
    sample = torch.randn(shp, generator=g).cuda()
    sig_table = torch.rand(5, 2, nb, generator=g).cuda()
    coef = torch.randn(5, 5, generator=g).cuda()
    noise = torch.randn((5,) + shp, generator=g).cuda()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    x_in, x_pre, sig = torch.zeros((nb,) + shp[1:], device="cuda"), torch.zeros((nb,) + shp[1:], device="cuda"), torch.zeros(nb, device="cuda")
    for st in range(3):
        ops.sampler_load(sample, x_in, x_pre, sig, sig_table, step, 1)
        assert torch.equal(x_in[:B], sample) and torch.equal(x_in[B:], sample) and torch.equal(x_pre, x_in) and torch.equal(sig, sig_table[st, 1])
        a, b = torch.randn(shp, generator=g).cuda(), torch.randn(shp, generator=g).cuda()
        want = ops.lincomb3(torch.empty_like(a), a, float(coef[st, 2]), b, float(coef[st, 3]), noise[st], float(coef[st, 4]))
        got = ops.lincomb3_dev(torch.empty_like(a), coef, step, a, 2, b, 3, noise, 4, z_step_stride=a.numel())
        assert torch.equal(got, want)
        want2 = ops.lincomb3(torch.empty_like(a), a, float(coef[st, 0]), b, float(coef[st, 1]))
        assert torch.equal(ops.lincomb3_dev(torch.empty_like(a), coef, step, a, 0, b, 1), want2)
        ops.step_advance(step)
    assert int(step.item()) == 3


def test_lincomb3():
    from dualdiffusion_amd import ops
    g = torch.Generator().manual_seed(0)
    x, y, z = (torch.randn(3, 5, 7, generator=g) for _ in range(3))
    out = torch.empty(3, 5, 7, device="cuda")
    ops.lincomb3(out, x.cuda(), 0.3, y.cuda(), -1.2, z.cuda(), 2.0)
    assert rel_l2(out, 0.3 * x - 1.2 * y + 2.0 * z) < 1e-6
    xc = x.cuda()
    ops.lincomb3(xc, xc, 0.5)
    assert rel_l2(xc, 0.5 * x) < 1e-7


def test_seamless_loop_matches_reference():
    """seamless_loop (per-step random roll + 32 wrapped columns, reference dual_diffusion_pipeline.py:651-658,729-732) with a reference
    input (x_ref), against the reference's own output."""
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.pipelines.dual_diffusion_pipeline import DualDiffusionPipeline, SampleParams
    t, m = load_golden("sampler")
    cfg = O.unet_cfg(**m["cfg"])
    unet = UNet(UNetConfig(**m["cfg"])).requires_grad_(False).train(False)
    unet.load_state_dict(O.random_unet_state(cfg, m["seed"]))
    unet = unet.to(device="cuda", dtype=torch.float32)

    class Fmt:
        ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)
    pipe = DualDiffusionPipeline({"unet": unet, "format": Fmt()})
    params = SampleParams(seed=4321, num_steps=3, batch_size=m["B"], sigma_max=m["sigma_max"], sigma_min=m["sigma_min"], sigma_data=1.0, rho=7.0,
                          schedule="edm2", seamless_loop=True, use_heun=True, cfg_scale=1.5, input_perturbation=1.0)
    out = pipe.diffusion_decode(params, quiet=True, audio_embedding=t["clap"], sample_shape=tuple(m["shape"]), x_ref=t["seamless.x_ref"],
                                noises=[t[f"seamless.noise{i}"] for i in range(3)])
    e = rel_l2(out, t["seamless.out"])
    print(f"sampler seamless_loop: rel-L2 {e:.3e}")
    assert e < 1e-4


def test_pipeline_from_pretrained_roundtrip(tmp_path):
    """save_pretrained -> model_index.json (reference package names) -> from_pretrained with checkpoint / EMA selection; the
    format-derived default sample shape of diffusion_decode (reference :230-300, :326-348, :617-622)."""
    import json
    import os
    from safetensors.torch import save_file
    from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.modules.vaes.vae_edm2 import AutoencoderKL_EDM2, DualDiffusionVAE_EDM2Config
    from dualdiffusion_amd.pipelines.dual_diffusion_pipeline import DualDiffusionPipeline, SampleParams
    ts, ms = load_golden("sampler")
    _tv, vm = load_golden("vae_small")
    ucfg, vcfg = O.unet_cfg(**ms["cfg"]), O.vae_cfg(**vm["cfg"])
    unet = UNet(UNetConfig(**ms["cfg"]))
    unet.load_state_dict(O.random_unet_state(ucfg, ms["seed"]))
    vae = AutoencoderKL_EDM2(DualDiffusionVAE_EDM2Config(**vm["cfg"]))
    vae.load_state_dict(O.random_vae_state(vcfg, vm["seed"]))
    fmt = SpectrogramFormat(SpectrogramFormatConfig())
    root = str(tmp_path / "model")
    DualDiffusionPipeline({"unet": unet, "vae": vae, "format": fmt}).save_pretrained(root)
    idx = json.load(open(os.path.join(root, "model_index.json")))
    assert idx["modules"]["unet"] == {"package": "dualdiffusion_amd.modules.unets.unet_edm2_b4", "class": "UNet"}
    # the reference spells the packages without our prefix: both load
    for d in idx["modules"].values():
        d["package"] = d["package"][len("dualdiffusion_amd."):]
    json.dump(idx, open(os.path.join(root, "model_index.json"), "w"))
    # a later checkpoint of the unet with an EMA file
    ck = os.path.join(root, "unet_checkpoint-200", "unet")
    os.makedirs(ck)
    unet.save_pretrained(ck)
    ema_sd = {k: v * 1.5 if v.ndim >= 2 else v for k, v in unet.state_dict().items()}      # un-normalised on purpose: load_ema re-normalises
    save_file({k: v.contiguous() for k, v in ema_sd.items()}, os.path.join(ck, "ema_0.9999.safetensors"))
    inv = DualDiffusionPipeline.get_model_module_inventory(root)
    assert inv["unet"].checkpoints == ["unet_checkpoint-200"] and inv["unet"].emas["unet_checkpoint-200"] == ["ema_0.9999.safetensors"]
    pipe = DualDiffusionPipeline.from_pretrained(root, torch_dtype=torch.float32, device="cuda", load_checkpoints=True, load_emas=True)
    assert type(pipe.unet).__name__ == "UNet" and pipe.unet.device.type == "cuda"
    assert pipe.model_metadata["load_emas"] == {"unet": "ema_0.9999.safetensors"}
    w = pipe.unet.state_dict()["enc.block0_layer0.conv_res0.weight"]
    assert rel_l2(w, O.rms_normalize(ema_sd["enc.block0_layer0.conv_res0.weight"])) < 1e-6
    # default sample shape: format -> vae latent shape -> unet latent shape
    mel = pipe.get_mel_spec_shape(bsz=2, raw_length=256 * 200)     # 201 frames -> cropped to 128
    lat = pipe.get_latent_shape(mel)
    assert tuple(mel) == (2, 2, 256, 128) and tuple(lat) == (2, 4, 64, 32), (mel, lat)
    params = SampleParams(seed=3, num_steps=2, batch_size=2, length=256 * 200, sigma_max=20.0, sigma_min=0.1, sigma_data=1.0)
    out = pipe.diffusion_decode(params, audio_embedding=ts["clap"])
    assert tuple(out.shape) == (2, 4, 64, 32) and torch.isfinite(out).all()
