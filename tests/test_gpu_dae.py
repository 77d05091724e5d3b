"""DAE_G1 on the HIP kernels (stereo depth folded into the image batch, axis-folded attention) against the reference module's outputs,
and the axis-folded attention op on its own against the oracle."""
import pytest
import torch

from oracle import dae_oracle as DO
from oracle import edm2_oracle as O
from tests.util import load_golden, rel_l2, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu


def _build(dtype):
    from dualdiffusion_amd.modules.daes.dae_edm2_g1 import DAE_G1, DAE_G1_Config
    t, m = load_golden("dae_g1_small")
    cfg = DO.dae_cfg(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in m["cfg"].items()})
    sd = DO.random_dae_state(cfg, m["seed"])
    dae = DAE_G1(DAE_G1_Config(**m["cfg"])).requires_grad_(False).train(False)
    assert set(dae.state_dict().keys()) == set(DO.dae_param_shapes(cfg).keys())
    dae.load_state_dict(sd, strict=True)
    return dae.to(device="cuda", dtype=dtype), t, m, cfg, sd


def test_dae_g1_fp32_vs_reference():
    dae, t, m, cfg, sd = _build(torch.float32)
    emb = dae.get_embeddings(t["emb_in"])
    assert rel_l2(emb, t["emb"]) < 1e-5
    lat = dae.encode(t["x"], emb)
    raw = dae.encode(t["x"], emb, normalize_latents=False)
    rec = dae.decode(t["latents"], emb)
    til = dae.tiled_encode(t["x_tiled"], emb[:1], **m["tiled"])
    errs = dict(encode=rel_l2(lat, t["latents"]), raw=rel_l2(raw, t["latents_raw"]), decode=rel_l2(rec, t["recon"]), tiled=rel_l2(til, t["latents_tiled"]))
    print("DAE_G1 fp32 vs reference: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert max(errs.values()) < 1e-4
    assert tuple(dae.get_latent_shape(t["x"].shape)) == tuple(t["latents"].shape) and tuple(dae.get_mel_spec_shape(t["latents"].shape)) == tuple(t["x"].shape)
    latents, recon, pre = dae(t["x"], emb)
    assert rel_l2(latents, t["latents"]) < 1e-4 and rel_l2(pre, t["latents_raw"]) < 1e-4 and rel_l2(recon, rec) < 1e-4
    # normalize_weights over dim 1 (MPConv3D_E.normalize_weights, dae_edm2_g1.py:123-126): a fixed point of the normalised state
    before = {k: v.clone() for k, v in dae.state_dict().items()}
    dae.normalize_weights()
    for k, v in dae.state_dict().items():
        if v.ndim > 2:
            assert rel_l2(v, O.rms_normalize(before[k].cpu(), dims=[1])) < 1e-5, k


def test_dae_g1_bf16():
    dae, t, m, cfg, sd = _build(torch.bfloat16)
    emb = dae.get_embeddings(t["emb_in"])
    lat = dae.encode(t["x"], emb)
    rec = dae.decode(t["latents"], emb)
    e1, e2 = rel_l2(lat, t["latents"]), rel_l2(rec, t["recon"])
    print(f"DAE_G1 bf16 vs fp32 reference: encode {e1:.2e}, decode {e2:.2e}")
    assert e1 < 3e-2 and e2 < 3e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_axis_folded_attention(dtype):
    """tokens along H for every (image, column) -- ddx_attn_fold_fwd on NHWC maps as they lie (reference dae_edm2_g1.py:209-228)."""
    from dualdiffusion_amd import ops
    g = torch.Generator().manual_seed(8)
    N, H, W, heads, d = 4, 37, 5, 2, 32
    Cn = heads * d
    rnd = lambda *s: torch.randn(*s, generator=g).to(dtype).float()      # noqa: E731
    qkv = rnd(N // 2, 3 * Cn, 2, H, W)                                    # reference layout (B, 3C, Z, H, W), rows head * 3d + dd * 3 + s
    ref = DO.axis_attention(qkv, heads)                                   # (B, C, Z, H, W)
    # what the module's prepared attn_qkv weights produce: [q|k (head, {q,k}, dd) | v (head, dd)] per image n = 2b + z
    idx = torch.arange(3 * Cn).view(heads, d, 3)
    rows = torch.cat([idx[:, :, :2].permute(0, 2, 1).reshape(-1), idx[:, :, 2].reshape(-1)])
    img = qkv[:, rows].permute(0, 2, 3, 4, 1).reshape(N, H, W, 3 * Cn).contiguous().to("cuda", dtype)
    out = ops.attention_fold(img[..., :2 * Cn], img[..., 2 * Cn:], heads)
    got = out.float().cpu().view(N // 2, 2, H, W, Cn).permute(0, 4, 1, 2, 3)
    e = rel_l2(got, ref)
    print(f"axis-folded attention {dtype}: {e:.3e}")
    assert e < (2e-5 if dtype == torch.float32 else 1e-2)
    # fold = 1 through the same entry is the plain attention over all pixels of an image
    a = ops.attention(img[..., :2 * Cn], img[..., 2 * Cn:], heads)
    full = O.attention_2d(_qk_full(qkv, heads, d), _v_full(qkv, heads, d), heads)
    assert rel_l2(a.float().cpu().view(N // 2, 2, H, W, Cn)[:, 0].permute(0, 3, 1, 2), full) < (2e-5 if dtype == torch.float32 else 1e-2)


def _qk_full(qkv, heads, d):
    """(B, 2C, H, W) in the reference UNet's attn_qk layout (head, dd, {q,k}) for depth slice 0."""
    B, C3, Z, H, W = qkv.shape
    x = qkv[:, :, 0].view(B, heads, d, 3, H, W)[:, :, :, :2]             # (B, heads, d, 2, H, W)
    return x.reshape(B, heads * d * 2, H, W)


def _v_full(qkv, heads, d):
    B, C3, Z, H, W = qkv.shape
    return qkv[:, :, 0].view(B, heads, d, 3, H, W)[:, :, :, 2].reshape(B, heads * d, H, W)


def test_latent_pre_encode_pipeline(tmp_path):
    """audio -> offset / mirror augmented crops -> raw_to_mel_spec -> dae.encode -> safetensors `latents[variation, C, h, w]` ->
    sliced read (reference dataset/processes/encode.py:306-353, training/dataset.py:192-201), against the same chain through the oracles."""
    import numpy as np
    from oracle import mel_oracle as M
    from dualdiffusion_amd.dataset.latents import EncodeProcessConfig, LatentPreEncoder, LatentsLoader, LatentsLoaderConfig, _normalize
    from dualdiffusion_amd.modules.formats.ms_mdct_dual import MS_MDCT_DualFormat, MS_MDCT_DualFormatConfig
    dae, t, m, cfg, sd = _build(torch.float32)
    fmt = MS_MDCT_DualFormat(MS_MDCT_DualFormatConfig(ms_width_alignment=16)).to(device="cuda")
    enc = LatentPreEncoder(fmt, dae, EncodeProcessConfig(latents_batch_size=1, latents_num_time_offset_augmentations=2,
                                                         latents_stereo_mirroring_augmentation=True))
    g = torch.Generator().manual_seed(3)
    audio = torch.randn(2, 256 * 40, generator=g) * 0.1
    clap = torch.randn(3, 32, generator=g)
    out = enc.encode(audio, clap)
    lat = out["latents"]
    # the reference sizes its batches from the number of OFFSETS (encode.py:262), so with mirroring on only the first
    # `offsets` crops of the interleaved [as is, mirrored, as is, mirrored, ...] list are encoded -- reproduced as it is
    assert lat.dtype == torch.bfloat16 and lat.shape[0] == 2 and lat.shape[1] == 8
    # the same chain through the oracles
    crop = fmt.get_raw_crop_width(audio.shape[-1] - enc.offset_padding)
    emb = DO.dae_embeddings(sd, _normalize(clap.mean(dim=0, keepdim=True)))
    for v, (off, flip) in enumerate([(0, False), (0, True)]):
        x = audio[:, off:off + crop].unsqueeze(0)
        if flip:
            x = torch.flip(x, dims=(1,))
        ref = DO.dae_encode(sd, cfg, M.raw_to_ms_mel_spec(x), emb)
        assert ref.shape[1:] == lat.shape[1:]
        assert rel_l2(lat[v].float(), ref[0]) < 6e-3, v                                      # (bf16 storage of the latents)
    path = str(tmp_path / "a" / "track.safetensors")
    LatentPreEncoder.save(path, out, {"prompt": "none"})
    got = LatentsLoader(LatentsLoaderConfig(latents_crop_width=lat.shape[-1] - 2, raw_crop_width=crop), rng=np.random.default_rng(0)).load(path)
    assert torch.equal(got["latents"], lat[got["variation"], ..., got["t_offset"]:got["t_offset"] + lat.shape[-1] - 2])
