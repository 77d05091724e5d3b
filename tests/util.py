"""Shared helpers for the tests: golden fixture loading and error metrics."""
import json
import os

import torch
from safetensors import safe_open

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    path = os.path.join(GOLD, name + ".safetensors")
    tensors = {}
    with safe_open(path, framework="pt") as f:
        meta = json.loads(f.metadata()["meta"])
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, meta


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def to_nhwc(x: torch.Tensor, dtype=None, device="cuda") -> torch.Tensor:
    y = x.permute(0, 2, 3, 1).contiguous()
    return y.to(device=device, dtype=dtype or x.dtype)


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous().float().cpu()
