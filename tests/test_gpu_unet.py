"""GPU parity of the whole UNet forward (HIP launch plan behind the reference module API) against the golden
vectors produced by the reference and against the CPU oracle.

float32: <= 1e-4 rel-L2 (BASELINE.json north_star tolerance; measured ~1e-6).
bfloat16 (bf16 storage, fp32 accumulate): <= 3e-2 vs the fp32 reference -- the reference's own bf16 forward is
1.2e-2 away from its fp32 forward on a tiny model.
"""
import pytest
import torch

from oracle import edm2_oracle as O
from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


class _Fmt:
    def __init__(self, fmin=20.0, fmax=16000.0):
        from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
        self.ms_freq_scale = FrequencyScale("mel", fmin, fmax, 32000, 3201, 256)


def _build(name, dtype, train_scale=False):
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    t, m = load_golden(name)
    cfg = O.unet_cfg(**m["cfg"])
    sd = {k[3:]: v for k, v in t.items() if k.startswith("sd.")} if m["weights"] == "stored" else O.random_unet_state(cfg, m["seed"])
    if train_scale:
        sd = {k: (v * t[f"trainscale.{k}"].view(-1, *([1] * (v.ndim - 1))) if v.ndim >= 2 else v) for k, v in sd.items()}
    unet = UNet(UNetConfig(**m["cfg"])).requires_grad_(False).train(False)
    missing = unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=dtype)
    return unet, t, m, cfg, sd


@pytest.mark.parametrize("name", ["unet_tiny", "unet_small", "unet_wide"])
def test_unet_forward_fp32(name):
    unet, t, m, cfg, sd = _build(name, torch.float32)
    fmt = _Fmt(*m["freq_range"])
    with torch.no_grad():
        emb = unet.get_embeddings(t["clap"], t["mask"].bool())
        assert rel_l2(emb, t["embeddings"]) < 1e-5
        assert rel_l2(unet.get_sigma_loss_logvar(t["sigma"].cuda()), t["logvar"]) < 1e-5
        out = unet(t["x_in"].cuda(), t["sigma"].cuda(), fmt, emb)
    e = rel_l2(out, t["out"])
    print(f"{name} fp32 forward rel-L2 vs reference golden: {e:.3e}")
    assert out.dtype == torch.float32 and e < 1e-4
    if "x_ref" in t:
        with torch.no_grad():
            out2 = unet(t["x_in"].cuda(), t["sigma"].cuda(), fmt, emb, x_ref=t["x_ref"].cuda(), perturbed_input=t["perturbed_input"].cuda())
        assert rel_l2(out2, t["out_xref"]) < 1e-4


@pytest.mark.parametrize("name", ["unet_small", "unet_wide"])
def test_unet_forward_bf16(name):
    unet, t, m, cfg, sd = _build(name, torch.bfloat16)
    fmt = _Fmt(*m["freq_range"])
    with torch.no_grad():
        emb = unet.get_embeddings(t["clap"], t["mask"].bool())
        out = unet(t["x_in"].cuda(), t["sigma"].cuda(), fmt, emb)
    e = rel_l2(out, t["out"])
    print(f"{name} bf16 forward rel-L2 vs fp32 reference golden: {e:.3e} (reference's own bf16-vs-fp32: ~1.2e-2)")
    assert e < 3e-2


def test_unet_train_mode_forward_fp32():
    """training=True applies the forced weight-norm inside every conv (mp_tools.py:360-361)."""
    unet, t, m, cfg, sd = _build("unet_small", torch.float32, train_scale=True)
    unet.train(True)
    fmt = _Fmt(*m["freq_range"])
    with torch.no_grad():
        out = unet(t["x_in"].cuda(), t["sigma"].cuda(), fmt, t["embeddings"].cuda())
    assert rel_l2(out, t["out_train_unnormalized"]) < 1e-4


def test_unet_graph_equals_eager_and_weight_refresh():
    unet, t, m, cfg, sd = _build("unet_wide", torch.bfloat16)
    fmt = _Fmt(*m["freq_range"])
    x, s, e = t["x_in"].cuda(), t["sigma"].cuda(), t["embeddings"].cuda()
    with torch.no_grad():
        a = unet(x, s, fmt, e)
        unet.compile()
        b = unet(x, s, fmt, e)
        c = unet(x, s, fmt, e)
        assert torch.equal(a, b) and torch.equal(b, c)
        # in-place weight change must be picked up (weight-prep cache keyed on parameter versions)
        unet.out_gain.mul_(2.0)
        d = unet(x, s, fmt, e)
    c_skip = (1 / (s ** 2 + 1)).view(-1, 1, 1, 1)
    assert rel_l2((d - c_skip * x), 2 * (a - c_skip * x)) < 2e-2


def test_unet_state_dict_keys_and_normalize_weights():
    unet, t, m, cfg, sd = _build("unet_small", torch.float32, train_scale=True)
    assert set(unet.state_dict().keys()) == set(O.unet_param_shapes(cfg).keys())
    unet.normalize_weights()
    got = unet.state_dict()
    for k in ("enc.block0_layer0.conv_res0.weight", "dec.block2_in0.attn_qk.weight", "emb_label.weight", "conv_out.weight"):
        assert rel_l2(got[k], O.rms_normalize(sd[k])) < 1e-6, k
    # logvar_linear has weight-norm disabled (unet_edm2_b4.py:187)
    assert rel_l2(got["logvar_linear.weight"], sd["logvar_linear.weight"]) < 1e-7


def test_vae_encode_decode_fp32_and_bf16():
    """AutoencoderKL_EDM2 on the HIP kernels against the reference's encode/decode (tests/golden/vae_small)."""
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.modules.vaes.vae_edm2 import AutoencoderKL_EDM2, DualDiffusionVAE_EDM2Config
    t, m = load_golden("vae_small")
    cfg = O.vae_cfg(**m["cfg"])
    sd = O.random_vae_state(cfg, m["seed"])

    class Fmt:
        fs = FrequencyScale("mel", *m["freq_range"], 32000, 3201, 256)

        def get_ln_freqs(self, x):
            ln = self.fs.get_unscaled(x.shape[2] + 2, device=x.device)[1:-1].log2()
            ln = ln.view(1, 1, -1, 1).repeat(x.shape[0], 1, 1, x.shape[3])
            return ((ln - ln.mean()) / ln.std()).to(x.dtype)

    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
        vae = AutoencoderKL_EDM2(DualDiffusionVAE_EDM2Config(**m["cfg"])).requires_grad_(False).train(False)
        assert set(vae.state_dict().keys()) == set(O.vae_param_shapes(cfg).keys())
        vae.load_state_dict(sd)
        vae = vae.to(device="cuda", dtype=dtype)
        with torch.no_grad():
            emb = vae.get_embeddings(t["labels_like"], labels_like=t["labels_like"])
            assert rel_l2(emb, t["emb"]) < (1e-5 if dtype == torch.float32 else 1e-2)
            dist = vae.encode(t["x"].cuda(), t["emb"].cuda(), Fmt())
            rec = vae.decode(t["latents"].cuda(), t["emb"].cuda(), Fmt())
        e1, e2 = rel_l2(dist.mode(), t["latents"]), rel_l2(rec, t["recon"])
        print(f"vae {dtype}: encode {e1:.3e} decode {e2:.3e}")
        assert e1 < tol and e2 < tol
        assert abs(float(dist.logvar) - float(t["noise_logvar"])) < 1e-6
        # batches above max_plan_batch run as chunks through one plan (other tile choices at the smaller batch: same tolerance;
        # the ln_freq table is still the whole batch's)
        vae.max_plan_batch = 1
        with torch.no_grad():
            dist1 = vae.encode(t["x"].cuda(), t["emb"].cuda(), Fmt())
            rec1 = vae.decode(t["latents"].cuda(), t["emb"].cuda(), Fmt())
        assert rel_l2(dist1.mode(), t["latents"]) < tol and rel_l2(rec1, t["recon"]) < tol
        assert tuple(vae.get_latent_shape(t["x"].shape)) == (2, 4, 8, 12) and tuple(vae.get_sample_shape((2, 4, 8, 12))) == (2, 2, 32, 48)


def test_resample2d():
    from dualdiffusion_amd import _lib as L
    from dualdiffusion_amd import ops
    from tests.util import to_nchw, to_nhwc
    x = torch.randn(2, 16, 6, 10, generator=torch.Generator().manual_seed(0))
    for dt in (torch.float32, torch.bfloat16):
        xr = x.to(dt).float()
        up = ops.resample2d(to_nhwc(xr, dt), torch.empty(2, 12, 20, 16, device="cuda", dtype=dt), L.RESAMPLE_UP)
        assert torch.equal(to_nchw(up), O.resample2x(xr, "up"))
        dn = ops.resample2d(to_nhwc(xr, dt), torch.empty(2, 3, 5, 16, device="cuda", dtype=dt), L.RESAMPLE_DOWN)
        assert rel_l2(to_nchw(dn), O.resample2x(xr, "down")) < (1e-6 if dt == torch.float32 else 4e-3)


def test_plan_two_lanes_match_single_lane():
    """ddx_plan_fork/main/join: ops recorded on the side lane give the same results eagerly and under graph capture."""
    import ctypes as C
    from dualdiffusion_amd import _lib as L, ops
    a = torch.randn(1 << 16, device="cuda")
    b = torch.randn(1 << 16, device="cuda")
    o1, o2, o3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    plan = L.Plan()
    with plan.record():
        ops.lincomb3(o1, a, 2.0, b, 1.0)                             # main
        L.check(L.lib().ddx_plan_fork(), "fork")
        ops.lincomb3(o2, o1, 1.0, a, -1.0)                           # side: needs o1
        L.check(L.lib().ddx_plan_main(), "main")
        ops.lincomb3(o3, o1, 0.5, b, 3.0)                            # main, concurrent with the side op
        L.check(L.lib().ddx_plan_join(), "join")
        ops.lincomb3(o1, o2, 1.0, o3, 1.0)                           # needs both
    ref = (a + b) + (0.5 * (2 * a + b) + 3 * b)
    plan.run()
    torch.cuda.synchronize()
    assert torch.allclose(o1, ref, atol=1e-5)
    o1.zero_()
    cap = torch.cuda.Stream()
    plan.graph_build(cap.cuda_stream)
    cap.synchronize()
    plan.graph_launch()
    torch.cuda.synchronize()
    assert torch.allclose(o1, ref, atol=1e-5)


def test_unet_odd_batches_and_ragged_shapes():
    """Tile selection / ragged edges: UNet forward at odd batch sizes and latent sizes that are not multiples of the kernel
    tiles, bf16 against the fp32 path of the same kernels (bf16 rounding only: rel-L2 <= 3e-2), outputs finite."""
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig

    class Fmt:
        ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)

    torch.manual_seed(0)
    cfg = UNetConfig(model_channels=64)
    sd, outs = None, {}
    shapes = [(3, 32, 176), (5, 16, 80), (2, 32, 48), (1, 48, 112)]
    for dt in (torch.float32, torch.bfloat16):
        unet = UNet(cfg).requires_grad_(False).train(False)
        if sd is None:
            sd = {k: v.clone() for k, v in unet.state_dict().items()}
            for k in sd:
                if sd[k].ndim == 0:
                    sd[k].fill_(0.7)
        unet.load_state_dict(sd)
        unet = unet.to(device="cuda", dtype=dt)
        unet.normalize_weights()
        for (B, H, W) in shapes:
            g = torch.Generator(device="cuda").manual_seed(B * 1000 + H + W)
            x = torch.randn(B, 4, H, W, device="cuda", generator=g)
            sigma = torch.exp(torch.randn(B, device="cuda", generator=g))
            clap = torch.randn(B, 512, device="cuda", generator=g)
            with torch.no_grad():
                emb = unet.get_embeddings(clap, torch.ones(B, dtype=torch.bool, device="cuda"))
                y = unet(x * (sigma.view(-1, 1, 1, 1) ** 2 + 1).sqrt(), sigma, Fmt(), emb)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (dt, B, H, W)
            outs[(dt, B, H, W)] = y.float()
    for (B, H, W) in shapes:
        e = rel_l2(outs[(torch.bfloat16, B, H, W)], outs[(torch.float32, B, H, W)])
        print(f"B={B} (4,{H},{W}): bf16 vs fp32 {e:.3e}")
        assert e < 3e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1.5e-2)], ids=["f32", "bf16"])
def test_up_block_skip_conv_at_source_size_matches_full_size(dtype, tol, monkeypatch):
    """engine.RES_UP: the 1x1 skip conv of an up block runs before the resample and conv_res1 gathers the half-size residual.
    Same dot products per pixel as conv_skip(upsample(x)): fp32 differs by summation order of a different kernel choice at most."""
    from dualdiffusion_amd import engine
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(engine, "RES_UP", flag)
        unet, t, m, cfg, sd = _build("unet_small", dtype)
        fmt = _Fmt(*m["freq_range"])
        with torch.no_grad():
            emb = unet.get_embeddings(t["clap"], t["mask"].bool())
            outs.append(unet(t["x_in"].cuda(), t["sigma"].cuda(), fmt, emb).float())
    e = rel_l2(outs[0], outs[1])
    print(f"RES_UP on vs off: {e:.3e}")
    assert e < tol


def test_qkv_twin_path_matches_the_prologue_path(monkeypatch):
    """Large batches: the merged attn_qk | attn_v conv reads a materialised x * c_qk twin (src0_alt) on the LDS-DMA kernel instead of
    scaling its operand in the register-staged kernel's prologue.  Default-size UNet at B=8 (2752 pixels at level 3), both ways."""
    from dualdiffusion_amd import engine
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    torch.manual_seed(3)
    cfg = UNetConfig()
    unet = UNet(cfg).requires_grad_(False).train(False)
    for k, v in unet.state_dict().items():
        if v.ndim == 0:
            v.fill_(0.7)
    sd = {k: v.clone() for k, v in unet.state_dict().items()}
    fmt = _Fmt()
    B = 8
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, 4, 32, 688, device="cuda", generator=g)
    sigma = torch.rand(B, device="cuda", generator=g) * 3 + 0.1
    clap = torch.randn(B, cfg.in_channels_emb, device="cuda", generator=g)
    outs, steps = [], []
    for min_px in (1, 0):
        monkeypatch.setattr(engine, "QKV_TWIN_MIN_PIXELS", min_px)
        u = UNet(cfg).requires_grad_(False).train(False)
        u.load_state_dict(sd)
        u = u.to(device="cuda", dtype=torch.bfloat16)
        u.normalize_weights()
        with torch.no_grad():
            emb = u.get_embeddings(clap, torch.ones(B, dtype=torch.bool, device="cuda"))
            outs.append(u(x, sigma, fmt, emb).float())
        steps.append(len(u._engine_for(B, 32, 688, False).pb.steps))
        del u
        torch.cuda.empty_cache()
    e = rel_l2(outs[0], outs[1])
    print(f"qkv twin path vs prologue path: rel-L2 {e:.3e}; plan steps {steps}")
    assert torch.isfinite(outs[0]).all() and e < 1.5e-2
    assert steps[0] >= steps[1]       # (the twin comes out of conv_res1's epilogue, or from one extra element-wise step per block)
