"""GPU parity of the MCLT diffusion-decoder UNet (DDec_MCLT_UNet_B1 on the 2-D conv kernels) against the reference fixture."""
import pytest
import torch

from oracle import ddec_oracle as DO
from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _build(dtype):
    from dualdiffusion_amd.modules.unets.unet_edm2_ddec_mclt_b1 import DDec_MCLT_UNet_B1, DDec_MCLT_UNet_B1_Config
    t, m = load_golden("ddec_small")
    cfg = DO.ddec_cfg(**m["cfg"])
    sd = DO.random_ddec_state(cfg, m["seed"])
    unet = DDec_MCLT_UNet_B1(DDec_MCLT_UNet_B1_Config(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in m["cfg"].items()}))
    unet.requires_grad_(False).train(False)
    unet.load_state_dict(sd, strict=True)
    return unet.to(device="cuda", dtype=dtype), t, m, cfg, sd


def test_ddec_forward_fp32():
    """fp32 kernels against the fp32 oracle (which the fixture pins to the reference through its bf16 twin)."""
    unet, t, m, cfg, sd = _build(torch.float32)
    out = unet(t["x_in"].cuda(), t["sigma"].cuda(), None, None, x_ref=t["x_ref"].cuda())
    e = rel_l2(out, t["out_fp32_oracle"])
    print(f"ddec fp32 forward rel-L2 vs fp32 oracle: {e:.3e}; vs the reference's bf16 forward: {rel_l2(out, t['out']):.3e}")
    assert out.shape == t["out"].shape and out.dtype == torch.float32 and e < 1e-4
    # perturbed_input path and logvar
    pert = t["x_in"] + 0.1 * torch.randn(t["x_in"].shape, generator=torch.Generator().manual_seed(2))
    ref2 = DO.ddec_forward(sd, cfg, t["x_in"], t["sigma"], t["x_ref"], perturbed_input=pert, compute_dtype=torch.float32)
    assert rel_l2(unet(t["x_in"].cuda(), t["sigma"].cuda(), None, None, x_ref=t["x_ref"].cuda(), perturbed_input=pert.cuda()), ref2) < 1e-4
    lv = DO.conv3d_mp(DO.fourier_mp(t["sigma"].log() / 4, sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"]), sd["logvar_linear.weight"])
    assert rel_l2(unet.get_sigma_loss_logvar(t["sigma"]).flatten(), lv.flatten()) < 1e-5


def test_ddec_forward_bf16_vs_reference():
    """bf16 (the reference hard-codes bfloat16 for this model's body) against the reference's own output."""
    unet, t, m, cfg, sd = _build(torch.bfloat16)
    out = unet(t["x_in"].cuda(), t["sigma"].cuda(), None, None, x_ref=t["x_ref"].cuda())
    e = rel_l2(out, t["out"])
    print(f"ddec bf16 forward rel-L2 vs reference (bf16): {e:.3e}")
    assert e < 3e-2


def test_ddec_requires_device_and_x_ref():
    from dualdiffusion_amd._lib import DDXError
    from dualdiffusion_amd.modules.unets.unet_edm2_ddec_mclt_b1 import DDec_MCLT_UNet_B1, DDec_MCLT_UNet_B1_Config
    unet = DDec_MCLT_UNet_B1(DDec_MCLT_UNet_B1_Config(in_num_freqs=32, in_psd_freqs=64, channel_mult=(1, 2), num_layers_per_block=1))
    with pytest.raises(DDXError):
        unet(torch.zeros(1, 2, 32, 8), torch.ones(1), None, None, x_ref=torch.zeros(1, 2, 64, 8))      # CPU: no fallback
    with pytest.raises(DDXError):
        unet.to("cuda")(torch.zeros(1, 2, 32, 8), torch.ones(1), None, None)                             # x_ref missing
