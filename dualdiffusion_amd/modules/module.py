"""Module base for the MI355X drop-in classes.

Mirrors the reference's module contract (reference src/modules/module.py:42-190) so that pipelines and
trainers written against it keep working: `{name}.json` dataclass config + `{name}.safetensors` weights
(`from_pretrained` / `save_pretrained`), `.to()/.half()` tracking of dtype/device/memory-format (half means
bfloat16, module.py:133-134), recursive `normalize_weights()`, and `compile()` -- which here builds a
hipGraph of the recorded launch plan instead of invoking a tracing compiler (module.py:145-149).
"""
from __future__ import annotations

import dataclasses
import inspect
import json
import os
from abc import ABC
from dataclasses import dataclass
from typing import Optional, Type, Union

import torch
from safetensors.torch import load_file, save_file


@dataclass
class DualDiffusionModuleConfig(ABC):
    last_global_step: int = 0


def config_from_dict(cls: Type, data: dict):
    """Tolerant dataclass construction: unknown keys are ignored (the reference only warns,
    src/utils/config.py:118-122), missing keys take the dataclass default."""
    names = {f.name for f in dataclasses.fields(cls)}
    return cls(**{k: v for k, v in data.items() if k in names})


def load_config(cls: Type, path: str):
    with open(path, "r") as f:
        return config_from_dict(cls, json.load(f))


def save_config(cfg, path: str) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump(dataclasses.asdict(cfg), f, indent=2)


_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
           "float16": torch.float16, "fp16": torch.float16}


class DualDiffusionModule(torch.nn.Module, ABC):

    config_class: Optional[Type[DualDiffusionModuleConfig]] = None
    module_name: Optional[str] = None
    has_trainable_parameters: bool = True
    supports_half_precision: bool = True
    supports_channels_last: Union[bool, str] = True
    supports_compile: bool = True

    def __init__(self) -> None:
        super().__init__()
        self.dtype = torch.get_default_dtype()
        self.device = torch.device("cpu")
        self.memory_format = torch.contiguous_format
        self.module_path = None

    # ------------------------------------------------------------------ persistence (module.py:59-99)
    @classmethod
    @torch.no_grad()
    def from_pretrained(cls, module_path: str, subfolder: Optional[str] = None, torch_dtype: Optional[torch.dtype] = None,
                        device: Optional[torch.device] = None, load_config_only: bool = False) -> "DualDiffusionModule":
        if subfolder is not None:
            module_path = os.path.join(module_path, subfolder)
        config_class = cls.config_class or inspect.signature(cls.__init__).parameters["config"].annotation
        name = os.path.basename(module_path)
        module = cls(load_config(config_class, os.path.join(module_path, f"{name}.json"))).requires_grad_(False).train(False)
        if (not load_config_only) and cls.has_trainable_parameters:
            module.load_state_dict(load_file(os.path.join(module_path, f"{name}.safetensors")))
        module.module_path = module_path
        return module.to(dtype=torch_dtype, device=device)

    @torch.no_grad()
    def save_pretrained(self, module_path: str, subfolder: Optional[str] = None, save_config_only: bool = False) -> None:
        if subfolder is not None:
            module_path = os.path.join(module_path, subfolder)
        os.makedirs(module_path, exist_ok=True)
        name = os.path.basename(module_path)
        save_config(self.config, os.path.join(module_path, f"{name}.json"))
        if type(self).has_trainable_parameters and not save_config_only:
            save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(module_path, f"{name}.safetensors"))

    # ------------------------------------------------------------------ placement (module.py:101-143)
    def to(self, device=None, dtype=None, memory_format=None, **kwargs) -> "DualDiffusionModule":
        if device is not None:
            device = torch.device(device)
        if dtype is not None:
            if isinstance(dtype, str):
                dtype = _DTYPES[dtype]
            if dtype in (torch.float16, torch.bfloat16) and not type(self).supports_half_precision:
                dtype = torch.float32
            if dtype == torch.float16:
                # the reference would keep fp16 parameters (module.py:107-111; only .half() means bfloat16, :133-134); the HIP
                # kernels compute in float32 or bfloat16, so fp16 is refused rather than silently changed
                raise ValueError("dualdiffusion_amd modules compute in float32 or bfloat16: float16 is not supported (use .half() / bfloat16)")
        if memory_format == torch.channels_last and not type(self).supports_channels_last:
            memory_format = None
        # parameters of this path are at most 4-D weights whose physical layout is re-done by weight preparation,
        # so memory_format only needs to be remembered, not applied
        super().to(device=device, dtype=dtype, **kwargs)
        self.dtype = dtype or self.dtype
        self.device = device or self.device
        self.memory_format = memory_format or self.memory_format
        self._on_placement_change()
        if getattr(self, "_normalize_on_placement", False) and self.device.type == "cuda":
            self._normalize_on_placement = False
            self.normalize_weights()
        return self

    def _on_placement_change(self) -> None:
        pass

    def float(self):
        return self.to(dtype=torch.float32)

    def half(self):
        return self.to(dtype=torch.bfloat16)

    def type(self, dtype):
        return self.to(dtype=dtype)

    def cpu(self, **kwargs):
        return self.to(device="cpu", **kwargs)

    def cuda(self, device: Optional[int] = None):
        return self.to(device="cuda" if device is None else f"cuda:{device}")

    def compile(self, **kwargs) -> None:
        """hipGraph capture of the launch plan replaces torch.compile (module.py:145-149)."""
        if type(self).supports_compile:
            self._use_graph = True

    @torch.no_grad()
    def load_ema(self, ema_path: str, phema_path: Optional[str] = None) -> None:
        if not os.path.isfile(ema_path):
            raise FileNotFoundError(f"Error: Could not find ema file '{ema_path}'")
        self.load_state_dict(load_file(ema_path))
        if self.device.type == "cuda":
            self.normalize_weights()
        else:   # the reference normalises on the host (module.py:173); here it is a HIP kernel: done when the module reaches the device
            self._normalize_on_placement = True

    @torch.no_grad()
    def normalize_weights(self) -> None:
        if not type(self).has_trainable_parameters:
            return
        for m in self.modules():
            if m is not self and hasattr(m, "normalize_weights"):
                m.normalize_weights()
