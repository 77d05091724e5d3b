"""dualdiffusion_amd: MI355X-native (gfx950) implementation of dualdiffusion's EDM2-UNet denoising / mel-latent hot path.

Host code is PyTorch-ROCm plumbing (device memory, streams, torch.distributed over RCCL); every hot op is a
hand-written HIP kernel behind the C ABI in include/ddx_hip.h (libddx_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
