"""torch.library registration of the module-level entry points (the reference compiles its modules with torch.compile,
src/modules/module.py:145-149, and wraps callers such as the sampling loop).

The HIP modules are launch plans over ctypes calls that Dynamo cannot see into: traced naively they graph-break at the first
ctypes call.  Each module forward is therefore ONE custom op with a fake (meta) implementation:

    dualdiffusion_amd::unet_forward(x_in, sigma, embeddings, x_ref?, perturbed_input?, module, format) -> float32 [B, C_out, H, W]
    dualdiffusion_amd::vae_encode(x, class_embeddings, module, format) -> latents mean
    dualdiffusion_amd::vae_decode(z, class_embeddings, module, format) -> sample

`module` / `format` are integer handles into a registry (custom ops take tensors and scalars only).  `UNet.forward`,
`AutoencoderKL_EDM2.encode / decode` route through these ops while `torch.compiler.is_compiling()`, so a compiled caller holds the
whole module call as one opaque node: no graph break, shapes and dtypes known to the tracer.  Inference only (the autograd bridge
of dualdiffusion_amd.autograd is a torch.autograd.Function, which Dynamo already traces as a unit).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from ._lib import DDXError

# handle -> (weak reference, type).  Dynamo traces `handle_of` itself (id() and a global dict store are supported), but the store is a
# DEFERRED side effect: while the caller is being traced the fake implementations see the handle before the registry does.  `_get` then
# looks the id up among the live, GC-tracked objects (the traced caller holds the object, so it is there) -- never by casting the integer
# to a pointer: a stale or foreign handle raises instead of being dereferenced.
# Lifetime: an entry dies WITH its object (weak-reference callback), so an id() that CPython hands to a new object later cannot resolve to
# the old slot, and nothing is kept alive by the registry.  Objects of types without weak references (none of the package's module /
# format classes) are held strongly until `release()`.
_OBJECTS: dict = {}


def handle_of(obj) -> int:
    """Integer handle of a module / format object."""
    h = id(obj)
    if torch.compiler.is_compiling():
        # traced by Dynamo: a plain (deferred) store and nothing that reads the registry -- a read would become a guard on the dict's
        # contents that the deferred store itself breaks.  `_get` upgrades the entry (death callback, type) the first time it resolves it.
        _OBJECTS[h] = (weakref.ref(obj), None)
        return h
    cur = _OBJECTS.get(h)
    if cur is not None and cur[1] is not None and cur[0]() is obj:
        return h
    try:
        _OBJECTS[h] = (weakref.ref(obj, lambda _r, h=h: _OBJECTS.pop(h, None)), type(obj))
    except TypeError:
        _OBJECTS[h] = ((lambda o=obj: o), type(obj))
    return h


def release(obj) -> None:
    """Drop the registry entry of `obj` (only needed for objects whose type has no weak references)."""
    _OBJECTS.pop(id(obj), None)


def _get(h: int):
    ent = _OBJECTS.get(h, None)
    obj = ent[0]() if ent is not None else None
    if obj is not None and ent[1] is None:                 # stored while tracing: give it the death callback and its type now
        handle_of(obj)
    elif obj is not None and type(obj) is not ent[1]:      # (cannot happen with the death callback; a foreign object under a recycled id)
        obj = None
    if obj is None:
        _OBJECTS.pop(h, None)
        # tracing: the handle precedes the registry store (see above).  One scan of the GC-tracked objects per unknown handle -- the hit is
        # registered, so a trace pays it once per module, not once per fake call
        import gc
        obj = next((o for o in gc.get_objects() if id(o) == h and (hasattr(o, "config") or hasattr(o, "ms_freq_scale") or hasattr(o, "_forward_plan"))), None)
        if obj is None:
            raise DDXError(f"dualdiffusion_amd custom op: unknown or expired module handle {h} (the module a compiled graph was traced "
                           "with must stay alive, and handles come from compile_ops.handle_of)")
        handle_of(obj)
    return obj


@torch.library.custom_op("dualdiffusion_amd::unet_forward", mutates_args=())
def unet_forward(x_in: torch.Tensor, sigma: torch.Tensor, embeddings: torch.Tensor, x_ref: Optional[torch.Tensor],
                 perturbed_input: Optional[torch.Tensor], module: int, format: int) -> torch.Tensor:
    return _get(module)._forward_plan(x_in, sigma, _get(format), embeddings, x_ref, perturbed_input)


@unet_forward.register_fake
def _(x_in, sigma, embeddings, x_ref, perturbed_input, module, format):
    cfg = _get(module).config
    return x_in.new_empty((x_in.shape[0], cfg.out_channels, x_in.shape[2], x_in.shape[3]), dtype=torch.float32)


@torch.library.custom_op("dualdiffusion_amd::vae_encode", mutates_args=())
def vae_encode(x: torch.Tensor, class_embeddings: torch.Tensor, module: int, format: int) -> torch.Tensor:
    return _get(module)._run_chunked("enc", x, class_embeddings, _get(format))


@vae_encode.register_fake
def _(x, class_embeddings, module, format):
    vae = _get(module)
    return x.new_empty(tuple(vae.get_latent_shape(x.shape)), dtype=torch.float32)


@torch.library.custom_op("dualdiffusion_amd::vae_decode", mutates_args=())
def vae_decode(z: torch.Tensor, class_embeddings: torch.Tensor, module: int, format: int) -> torch.Tensor:
    return _get(module)._run_chunked("dec", z, class_embeddings, _get(format))


@vae_decode.register_fake
def _(z, class_embeddings, module, format):
    vae = _get(module)
    return z.new_empty(tuple(vae.get_sample_shape(z.shape)), dtype=torch.float32)
