"""Latents file format and the slicing reader (CPU): what dataset/latents.LatentsLoader returns for a pre-encoded track against the
lines of the reference's DatasetTransform (training/dataset.py:192-236) restated inline with the same numpy draws."""
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
    sel = emb[si:ei + 1].float().clone()
    if sf > 0:
        sel[0] = _normalize(_mp_sum(emb[si], emb[si + 1], sf).unsqueeze(0))[0]
    if ef > 0:
        sel[-1] = _normalize(_mp_sum(emb[ei], emb[ei + 1], ef).unsqueeze(0))[0]
    ref = _normalize(sel.sum(dim=0).unsqueeze(0))[0]
    assert torch.allclose(out["audio_embeddings"], ref, atol=1e-6)
    b = LatentsLoader(cfg, rng=np.random.default_rng(1)).batch([path, path, path])
    assert b["latents"].shape == (3, 8, 32, 688) and b["audio_embeddings"].shape == (3, 512)
