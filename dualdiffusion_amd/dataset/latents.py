"""Latent pre-encode and the latents loader: the step that feeds the training hot path.

Device part of reference src/dataset/processes/encode.py:212-366 (`EncodeProcess.process`): raw audio -> time-offset (and stereo
mirror / pitch) augmented crops -> mel spectrograms (`format.raw_to_mel_spec`, HIP) -> `dae.(tiled_)encode` (HIP) in batches ->
`latents[variation, C, H, W]` (bfloat16) stored next to `clap_audio_embeddings` in one safetensors file per track; and the reader
of reference src/training/dataset.py:192-236 (`DatasetTransform.__call__`): a random variation and a random time crop read as a
SLICE of the file (no full load), plus the crop's averaged CLAP audio embedding with the reference's fractional end-point blend.
The CLAP encoder itself (modules/embeddings) is outside the hot path: embeddings are an input here.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
from safetensors import safe_open
from safetensors.torch import save_file


@dataclass
class EncodeProcessConfig:
    """The latent-related fields of the reference's EncodeProcessConfig (encode.py:44-63)."""
    latents_batch_size: int = 1
    latents_num_time_offset_augmentations: int = 8
    latents_pitch_offset_augmentations: list = field(default_factory=list)
    latents_stereo_mirroring_augmentation: bool = True
    latents_tiled_encode: bool = False
    latents_tiled_max_chunk_size: int = 6144
    latents_tiled_overlap: int = 256


def _normalize(x: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """mp_tools.normalize over all dims but 0 (tiny [n, 512] CLAP rows: host arithmetic)."""
    n = torch.linalg.vector_norm(x.float(), dim=list(range(1, x.ndim)), keepdim=True)
    n = torch.add(eps, n, alpha=math.sqrt(n.numel() / x.numel()))
    return (x.float() / n).to(x.dtype)


def _mp_sum(a: torch.Tensor, b: torch.Tensor, t: float) -> torch.Tensor:
    """mp_tools.mp_sum with a python-float t (mp_tools.py:274-279): in the operands' dtype -- the stored CLAP rows are bfloat16 and the
    reference's loader mixes and averages them in bfloat16 (training/dataset.py:228-236)."""
    return a.lerp(b, t) / ((1 - t) ** 2 + t ** 2) ** 0.5


class LatentPreEncoder:
    """format + dae on the ROCm device -> `encode(audio, clap_audio_embeddings)` -> tensors for the per-track safetensors file."""

    def __init__(self, format, dae, config: EncodeProcessConfig = EncodeProcessConfig()) -> None:
        self.format, self.dae, self.cfg = format, dae, config
        if config.latents_pitch_offset_augmentations:
            raise NotImplementedError("pitch-offset augmentation formats (encode.py:224-229) are not built")
        n = config.latents_num_time_offset_augmentations
        hop = format.config.ms_frame_hop_length
        self.offsets = [i * hop for i in range(n)]                    # encode.py:258-260
        self.offset_padding = hop * n if n > 0 else 0
        self.num_batches_per_sample = (n + config.latents_batch_size - 1) // config.latents_batch_size

    @torch.no_grad()
    def encode(self, audio: torch.Tensor, clap_audio_embeddings: torch.Tensor) -> dict:
        """audio [C, L] at the format's sample rate; clap_audio_embeddings [n, 512].  Returns {"latents": [variations, C, h, w] bf16,
        "clap_audio_embeddings": bf16} (encode.py:306-353)."""
        cfg, fmt, dae = self.cfg, self.format, self.dae
        dev = dae.device
        audio = audio.to(dev, torch.float32)
        crop_width = fmt.get_raw_crop_width(audio.shape[-1] - self.offset_padding)
        crops = []
        for off in self.offsets:
            x = audio[:, off:off + crop_width].unsqueeze(0)
            crops.append(x)
            if cfg.latents_stereo_mirroring_augmentation:
                crops.append(torch.flip(x, dims=(1,)))
        crops = torch.cat(crops, dim=0).contiguous()
        bsz = cfg.latents_batch_size
        # the reference hands the DAE a bfloat16 mel spectrogram (encode.py:329 `.type(torch.bfloat16)`): round here too, so that
        # latents encoded by an fp32 DAE are comparable with reference-encoded datasets
        mels = [fmt.raw_to_mel_spec(crops[b * bsz:(b + 1) * bsz]).to(torch.bfloat16).float() for b in range(self.num_batches_per_sample)]
        mel = torch.cat(mels, dim=0)
        emb = _normalize(clap_audio_embeddings.float().mean(dim=0, keepdim=True)).to(dev)
        dae_emb = dae.get_embeddings(emb)
        out = []
        for b in range(mel.shape[0] // bsz):
            m = mel[b * bsz:(b + 1) * bsz]
            e = dae_emb.expand(m.shape[0], -1) if dae_emb is not None else None
            if cfg.latents_tiled_encode:
                out.append(dae.tiled_encode(m, e, max_chunk=cfg.latents_tiled_max_chunk_size, overlap=cfg.latents_tiled_overlap))
            else:
                out.append(dae.encode(m, e))
        latents = torch.cat(out, dim=0).to(torch.bfloat16)
        assert latents.ndim == 4
        return {"latents": latents.cpu(), "clap_audio_embeddings": clap_audio_embeddings.to(torch.bfloat16).cpu()}

    @staticmethod
    def save(path: str, tensors: dict, metadata: Optional[dict] = None) -> None:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata={k: str(v) for k, v in (metadata or {}).items()})


@dataclass
class LatentsLoaderConfig:
    latents_crop_width: int = 688
    raw_crop_width: int = 1408768
    sample_rate: int = 32000
    audio_embedding_duration: float = 10.0       # seconds of audio per CLAP embedding row (clap_config.audio_embedding_duration)


class LatentsLoader:
    """Random (variation, time crop) of pre-encoded latents + the crop's CLAP audio embedding (training/dataset.py:192-236).
    `rng`: numpy Generator or RandomState-like with `integers` / `randint`."""

    def __init__(self, config: LatentsLoaderConfig = LatentsLoaderConfig(), rng=None) -> None:
        self.cfg = config
        self.rng = rng if rng is not None else np.random.default_rng()

    def _randint(self, lo: int, hi: int) -> int:
        return int(self.rng.integers(lo, hi)) if hasattr(self.rng, "integers") else int(self.rng.randint(lo, hi))

    def load(self, latents_file_name: str) -> dict:
        c = self.cfg
        with safe_open(latents_file_name, framework="pt") as f:
            sl = f.get_slice("latents")
            shape = sl.get_shape()
            idx = self._randint(0, shape[0])                                              # random variation
            t0 = self._randint(0, shape[-1] - c.latents_crop_width + 1)                   # random time offset
            latents = sl[idx, ..., t0:t0 + c.latents_crop_width]                          # partial read
            emb_sl = f.get_slice("clap_audio_embeddings")
            emb_len = emb_sl.get_shape()[0]
            sec_per_px = c.raw_crop_width / c.sample_rate / c.latents_crop_width
            start = t0 * sec_per_px / c.audio_embedding_duration
            end = (t0 + c.latents_crop_width) * sec_per_px / c.audio_embedding_duration
            start = float(np.clip(start - 0.5, 0, emb_len - 1))
            end = float(np.clip(end - 0.5, start, emb_len - 1))
            s_int, s_frac, e_int, e_frac = int(start), start % 1, int(end), end % 1
            selected = emb_sl[s_int:e_int + 1]            # stored dtype (bfloat16), as the reference
            if s_frac > 0:
                selected[0] = _normalize(_mp_sum(emb_sl[s_int], emb_sl[s_int + 1], s_frac).unsqueeze(0))[0]
            if e_frac > 0:
                selected[-1] = _normalize(_mp_sum(emb_sl[e_int], emb_sl[e_int + 1], e_frac).unsqueeze(0))[0]
            audio_emb = _normalize(selected.sum(dim=0).unsqueeze(0))[0]
        return {"latents": latents, "audio_embeddings": audio_emb, "variation": idx, "t_offset": t0}

    def batch(self, files: list) -> dict:
        items = [self.load(p) for p in files]
        return {"sample_paths": list(files), "latents": torch.stack([i["latents"] for i in items]),
                "audio_embeddings": torch.stack([i["audio_embeddings"] for i in items])}
