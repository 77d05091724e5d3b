"""CPU restatement of the reference's 2-D multi-scale spectral loss (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/training/loss/multiscale_spectral.py:121-296 (`MSSLoss2DConfig`, `MSSLoss2D`): per block
width w (step = max(w // block_overlap, 1)): reflect-pad by w/2, unfold into w x w blocks, multiply by the unit-RMS 2-D
window, `rfft2(norm="ortho")`, optional mid/side "stack" / "cat", weighted L1 (or MSE) between the magnitudes (and, with phase_loss_scale, the
real / imaginary parts), static or target-derived ("dynamic") frequency weights, mean over
everything but the batch, summed over the widths.  Pinned against the reference by tools/make_golden.py
(tests/golden/mss_loss.safetensors); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math

import torch


def flat_top_1d(w: int) -> torch.Tensor:
    """multiscale_spectral.py:176-178,182-184 (x = 2*pi*n/w, no half-sample offset)."""
    x = torch.arange(w) / w * 2 * torch.pi
    return (0.21557895 - 0.41663158 * torch.cos(x) + 0.277263158 * torch.cos(2 * x)
            - 0.083578947 * torch.cos(3 * x) + 0.006947368 * torch.cos(4 * x))


def block_window(w: int, fn: str = "flat_top") -> torch.Tensor:
    """2-D block window normalised to unit RMS (multiscale_spectral.py:147-165)."""
    if fn == "flat_top":
        w1 = flat_top_1d(w)
        win = w1.view(-1, 1) * w1.view(1, -1)
    elif fn == "hann":
        w1 = (torch.arange(w) / w * torch.pi).sin() ** 2
        win = w1.view(-1, 1) * w1.view(1, -1)
    elif fn == "kaiser":
        w1 = torch.kaiser_window(w, beta=12, periodic=False)
        win = torch.outer(w1, w1)
    elif fn == "none":
        win = torch.ones(w, w)
    elif fn == "flat_top_circular":
        # multiscale_spectral.py:200-211: radial flat-top over the distance from the block centre (pixel centres), zero outside the circle
        xc = (torch.arange(w) + 0.5).view(1, -1)
        yc = (torch.arange(w) + 0.5).view(-1, 1)
        dist = torch.sqrt((xc - w / 2) ** 2 + (yc - w / 2) ** 2) / (w // 2)
        x = dist * torch.pi + torch.pi
        win = (0.21557895 - 0.41663158 * torch.cos(x) + 0.277263158 * torch.cos(2 * x)
               - 0.083578947 * torch.cos(3 * x) + 0.006947368 * torch.cos(4 * x)) * (dist <= 1)
    else:
        raise ValueError(fn)
    return win / win.square().mean().sqrt()


def loss_weight(w: int, weighting: str = "product", exponent: float = 1.0, width_exponent: float = 0.0) -> torch.Tensor:
    """Static frequency weights [w, w//2+1] (multiscale_spectral.py:167-174, 255-259)."""
    fh = torch.fft.fftfreq(w, d=1 / w)
    fw = torch.fft.rfftfreq(w, d=1 / w)
    if weighting == "product":
        lw = (fh.view(-1, 1).abs() + 1) * (fw.view(1, -1).abs() + 1)
    elif weighting == "f^2":
        lw = fh.view(-1, 1) ** 2 + fw.view(1, -1) ** 2 + 1
    else:
        raise ValueError(weighting)
    lw = lw.float()
    if exponent != 1:
        lw = lw.pow(exponent)
    if width_exponent != 0:
        lw = lw * (w ** width_exponent)
    return lw


def stft2d(x: torch.Tensor, w: int, step: int, window: torch.Tensor, midside: str = "stack") -> torch.Tensor:
    """multiscale_spectral.py:213-235."""
    pad = w // 2
    x = torch.nn.functional.pad(x, (pad, pad, pad, pad), mode="reflect")
    x = x.unfold(2, w, step).unfold(3, w, step)
    x = x * window
    x = torch.fft.rfft2(x, norm="ortho")
    if midside == "stack":
        x = torch.stack((x[:, 0] + x[:, 1], x[:, 0] - x[:, 1]), dim=1)
    elif midside == "cat":      # :229-231: (L, R, (L+R)/sqrt 2, (L-R)/sqrt 2)
        x = torch.cat((x, (x[:, 0:1] + x[:, 1:2]) * 0.5 ** 0.5, (x[:, 0:1] - x[:, 1:2]) * 0.5 ** 0.5), dim=1)
    elif midside != "none":
        raise ValueError(midside)
    return x


def mss_loss(sample: torch.Tensor, target: torch.Tensor, block_widths=(8, 16, 32, 64), block_overlap: int = 8,
             window_fn: str = "flat_top", weighting: str = "product", weight_exponent: float = 1.0,
             width_weight_exponent: float = 0.0, midside: str = "stack", use_mse: bool = False,
             abs_loss_scale: float = 1.0, phase_loss_scale: float = 0.0) -> torch.Tensor:
    """multiscale_spectral.py:237-294.  Returns the per-sample loss [B]."""
    loss = torch.zeros(target.shape[0], dtype=sample.dtype)
    for w in block_widths:
        if w > target.shape[-1]:
            continue
        step = max(w // block_overlap, 1)
        win = block_window(w, window_fn).to(sample.dtype)
        with torch.no_grad():
            t_fft = stft2d(target, w, step, win, midside)
            t_abs = t_fft.abs()
            if weighting == "dynamic":      # :252-253: per channel and frequency, from the target's mean magnitude over batch and blocks
                lw = 1 / t_abs.mean(dim=(0, 2, 3), keepdim=True).clip(min=1e-2)
                if weight_exponent != 1:
                    lw = lw.pow(weight_exponent)
                if width_weight_exponent != 0:
                    lw = lw * (w ** width_weight_exponent)
            else:
                lw = loss_weight(w, weighting, weight_exponent, width_weight_exponent).to(sample.dtype)
        s_fft = stft2d(sample, w, step, win, midside)
        block = torch.zeros_like(t_abs)
        if abs_loss_scale > 0:
            d = s_fft.abs() - t_abs
            block = (d * d if use_mse else d.abs()) * abs_loss_scale
        if phase_loss_scale > 0:            # :275-277 / :286-288: the same distance on the real and imaginary parts
            dr, di = s_fft.real - t_fft.real, s_fft.imag - t_fft.imag
            block = block + ((dr * dr + di * di) if use_mse else (dr.abs() + di.abs())) * phase_loss_scale
        loss = loss + (block * lw).mean(dim=(1, 2, 3, 4, 5))
    return loss


def mss_loss_and_grad(sample: torch.Tensor, target: torch.Tensor, **kw):
    """Loss [B] and d(sum_b loss_b)/d(sample) via autograd (the gradient a trainer's backward() delivers)."""
    s = sample.detach().clone().requires_grad_(True)
    loss = mss_loss(s, target, **kw)
    loss.sum().backward()
    return loss.detach(), s.grad.detach()
