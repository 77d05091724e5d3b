"""Latents file format and the slicing reader (CPU): what dataset/latents.LatentsLoader returns for a pre-encoded track against the
OUTPUT of the reference's own DatasetTransform run on the same file with the same numpy draws (tests/golden/latents_loader, generated
by tools/make_golden.py gen_loader), and against the transform's lines restated inline."""
import numpy as np
import torch

from dualdiffusion_amd.dataset.latents import LatentPreEncoder, LatentsLoader, LatentsLoaderConfig, _mp_sum, _normalize


def test_latents_loader_slices_like_the_reference(tmp_path):
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(6, 8, 32, 1000, generator=g).to(torch.bfloat16)
    emb = _normalize(torch.randn(7, 512, generator=g)).to(torch.bfloat16)
    path = str(tmp_path / "track.safetensors")
    LatentPreEncoder.save(path, {"latents": lat, "clap_audio_embeddings": emb}, {"prompt": "x"})
    cfg = LatentsLoaderConfig(latents_crop_width=688, raw_crop_width=1408768, sample_rate=32000, audio_embedding_duration=10.0)
    out = LatentsLoader(cfg, rng=np.random.RandomState(5)).load(path)
    # the reference's draws: np.random.randint(0, variations), np.random.randint(0, W - crop + 1)
    r = np.random.RandomState(5)
    idx, t0 = r.randint(0, 6), r.randint(0, 1000 - 688 + 1)
    assert out["variation"] == idx and out["t_offset"] == t0
    assert torch.equal(out["latents"], lat[idx, ..., t0:t0 + 688]) and out["latents"].dtype == torch.bfloat16
    sec_per_px = 1408768 / 32000 / 688
    a0, a1 = t0 * sec_per_px / 10.0, (t0 + 688) * sec_per_px / 10.0
    start = np.clip(a0 - 0.5, 0, 6)
    end = np.clip(a1 - 0.5, start, 6)
    si, sf, ei, ef = int(start), start % 1, int(end), end % 1
    sel = emb[si:ei + 1].clone()          # bfloat16 rows: the reference mixes / sums them in the stored dtype
    if sf > 0:
        sel[0] = _normalize(_mp_sum(emb[si], emb[si + 1], sf).unsqueeze(0))[0]
    if ef > 0:
        sel[-1] = _normalize(_mp_sum(emb[ei], emb[ei + 1], ef).unsqueeze(0))[0]
    ref = _normalize(sel.sum(dim=0).unsqueeze(0))[0]
    assert torch.equal(out["audio_embeddings"], ref) and ref.dtype == torch.bfloat16
    b = LatentsLoader(cfg, rng=np.random.default_rng(1)).batch([path, path, path])
    assert b["latents"].shape == (3, 8, 32, 688) and b["audio_embeddings"].shape == (3, 512)


def test_latents_loader_vs_reference_transform_output(tmp_path):
    """Bit-for-bit the reference's DatasetTransform.__call__ outputs (latents crops and CLAP crop averages of four draws)."""
    from tests.util import load_golden
    t, m = load_golden("latents_loader")
    path = str(tmp_path / "track.safetensors")
    LatentPreEncoder.save(path, {"latents": t["latents"], "clap_audio_embeddings": t["clap_audio_embeddings"]})
    cfg = LatentsLoaderConfig(latents_crop_width=m["latents_crop_width"], raw_crop_width=m["raw_crop_width"], sample_rate=m["sample_rate"],
                              audio_embedding_duration=m["audio_embedding_duration"])
    b = LatentsLoader(cfg, rng=np.random.RandomState(m["numpy_seed"])).batch([path] * m["n"])
    assert torch.equal(b["latents"], t["out_latents"])
    assert b["audio_embeddings"].dtype == t["out_audio_embeddings"].dtype == torch.bfloat16
    assert torch.equal(b["audio_embeddings"], t["out_audio_embeddings"])
