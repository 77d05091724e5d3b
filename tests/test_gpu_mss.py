"""GPU parity of the fused multi-scale spectral loss kernel (value + gradient) against the reference's MSSLoss2D fixtures
and, at the full mel-spectrogram size, through size-independent properties."""
import pytest
import torch

from tests.util import load_golden, rel_l2

pytestmark = pytest.mark.gpu

# fp32 block FFTs in LDS against the reference's fp32 rfft2; the gradient is a sum of 64 overlapping block contributions
# accumulated with float atomics (order varies run to run)
TOL_LOSS, TOL_GRAD = 2e-5, 1e-4


def _loss(meta):
    from dualdiffusion_amd.training.loss.multiscale_spectral import MSSLoss2D, MSSLoss2DConfig
    cfg = MSSLoss2DConfig(block_widths=tuple(meta["block_widths"]), block_overlap=meta["block_overlap"], block_window_fn=meta["window_fn"],
                          frequency_weighting=meta["weighting"], frequency_weight_exponent=meta["weight_exponent"],
                          block_width_weight_exponent=meta["width_weight_exponent"], use_midside_transform=meta["midside"],
                          use_mse_loss=meta["use_mse"], abs_loss_scale=meta.get("abs_loss_scale", 1.0), phase_loss_scale=meta.get("phase_loss_scale", 0.0))
    return MSSLoss2D(cfg, torch.device("cuda"))


# (the last five: every MSSLoss2DConfig option beyond the defaults -- circular window, mid/side "cat", phase terms with L1 and MSE, dynamic
# frequency weights from the statistics launch, and their combination)
@pytest.mark.parametrize("name", ["default", "hann_f2_mse", "ragged", "circular_cat", "phase_l1", "phase_mse_cat", "dynamic", "dynamic_cat_phase"])
def test_mss_matches_reference(name):
    t, m = load_golden("mss_loss")
    mss = _loss(m[name])
    sample = t[f"{name}.sample"].cuda().requires_grad_(True)
    loss = mss.mss_loss(sample, t[f"{name}.target"].cuda())
    loss.sum().backward()
    el, eg = rel_l2(loss.detach(), t[f"{name}.loss"]), rel_l2(sample.grad, t[f"{name}.grad"])
    print(f"mss {name}: loss rel-L2 {el:.3e}, grad rel-L2 {eg:.3e}")
    assert el < TOL_LOSS and eg < TOL_GRAD, (el, eg)


def test_mss_autograd_weighting_and_value_only():
    """backward() scales the stored gradient by the incoming per-sample weights; no gradient buffer without requires_grad."""
    t, m = load_golden("mss_loss")
    mss = _loss(m["default"])
    s, tg = t["default.sample"].cuda(), t["default.target"].cuda()
    loss0 = mss.mss_loss(s, tg)                      # value only
    assert not loss0.requires_grad and rel_l2(loss0, t["default.loss"]) < TOL_LOSS
    s2 = s.clone().requires_grad_(True)
    wts = torch.tensor([0.25, -2.0], device="cuda")
    (mss.mss_loss(s2, tg) * wts).sum().backward()
    assert rel_l2(s2.grad, t["default.grad"] * wts.cpu().view(-1, 1, 1, 1)) < TOL_GRAD


def test_mss_full_size_properties():
    """Full mel-spectrogram size (2, 256, 5504): identical inputs give zero loss, the loss is symmetric in (sample, target),
    scaling both by a scales the loss by a, and the analytic gradient matches a directional finite difference."""
    from dualdiffusion_amd.training.loss.multiscale_spectral import MSSLoss2D, MSSLoss2DConfig
    mss = MSSLoss2D(MSSLoss2DConfig(), torch.device("cuda"))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 2, 256, 5504, generator=g).cuda()
    y = (x * 0.8 + 0.3 * torch.randn(1, 2, 256, 5504, generator=g).cuda())
    l_xx, g_xx = mss.mss_loss_and_grad(x, x)
    assert float(l_xx.abs().max()) < 1e-6
    l_xy, g_xy = mss.mss_loss_and_grad(x, y)
    l_yx, _ = mss.mss_loss_and_grad(y, x)
    assert abs(float(l_xy - l_yx)) < 2e-5 * float(l_xy)
    l_2, _ = mss.mss_loss_and_grad(2 * x, 2 * y)
    assert abs(float(l_2 - 2 * l_xy)) < 2e-5 * float(l_2)
    d = torch.randn(x.shape, generator=g).cuda()
    eps = 5e-2
    lp, _ = mss.mss_loss_and_grad(x + eps * d, y)
    lm, _ = mss.mss_loss_and_grad(x - eps * d, y)
    fd = float(lp - lm) / (2 * eps)
    an = float((g_xy * d).sum())
    print(f"mss full size: loss {float(l_xy):.4f}, directional derivative fd {fd:.5f} vs analytic {an:.5f}")
    assert abs(fd - an) < 5e-2 * abs(an) + 1e-4   # central difference of an fp32 L1-type loss: kinks + rounding noise
