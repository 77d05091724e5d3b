"""Import harness for the read-only reference at /root/reference (BUILD CONTAINER ONLY).

Our own tooling.  It never ships reference code: it only makes the
reference importable in this container so that `tools/make_golden.py` can emit golden
input/output vectors into `tests/golden/`.  Nothing here is used on the GPU box.

The reference needs a handful of third-party packages that are absent from the image and do
no arithmetic on the hot path (dotenv, pyjson5, torchaudio, cv2, mutagen, pyloudnorm,
librosa); they are replaced by empty stub modules.  `torchaudio.transforms.Spectrogram` is
the one exception: it is a thin wrapper over `torch.stft` (documented semantics), supplied
below so that the reference's mel-STFT format constructs and runs.
"""
from __future__ import annotations

import importlib.machinery
import json
import sys
import types

import torch

REF_SRC = "/root/reference/src"


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Spectrogram(torch.nn.Module):
    """torch.stft-backed stand-in for torchaudio.transforms.Spectrogram (documented behaviour)."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window,
                 power=2.0, normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 2
        self.pad = pad
        self.power = power
        self.normalized = normalized
        self.center = center
        self.pad_mode = pad_mode
        self.onesided = onesided
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window, persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.pad > 0:
            x = torch.nn.functional.pad(x, (self.pad, self.pad))
        shape = x.shape
        y = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window,
                       self.center, self.pad_mode, False, self.onesided, return_complex=True)
        y = y.reshape(shape[:-1] + y.shape[-2:])
        if self.normalized in (True, "window"):
            y = y / self.window.pow(2.0).sum().sqrt()
        if self.power is not None:
            y = y.abs() if self.power == 1.0 else y.abs().pow(self.power)
        return y


_installed = False


def install() -> None:
    """Make `modules.*`, `training.*`, `sampling.*`, `pipelines.*` of the reference importable."""
    global _installed
    if _installed:
        return
    _stub("dotenv", load_dotenv=lambda *a, **k: None)
    _stub("pyjson5", load=json.load, loads=json.loads)
    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms", Spectrogram=_Spectrogram)
    ta.functional = _stub("torchaudio.functional")
    _stub("cv2", IMREAD_UNCHANGED=-1)
    mg = _stub("mutagen")
    mg.flac = _stub("mutagen.flac")
    _stub("pyloudnorm")
    _stub("librosa")
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    _installed = True


def install_old_vae() -> None:
    """Alias the stale `modules.vaes` import path used by the old VAE (old/vaes/vae_edm2.py:30)."""
    install()
    import importlib
    old_vaes = importlib.import_module("modules.old.vaes")
    sys.modules["modules.vaes"] = old_vaes
    sys.modules["modules.vaes.vae"] = importlib.import_module("modules.old.vaes.vae")
