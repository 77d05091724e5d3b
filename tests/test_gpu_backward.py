"""One EDM2 block forward + backward on the HIP kernels (dualdiffusion_amd.training.block_grad) against torch autograd
through the oracle's definitions (training mode: forced weight norm inside the forward)."""
import pytest
import torch

from oracle import edm2_oracle as O
from tests.util import rel_l2, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu


def _r(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("case", ["cat_skip", "plain"])
def test_block_forward_backward(case):
    from dualdiffusion_amd.training.block_grad import block_backward, block_forward_train
    g = torch.Generator().manual_seed(7 if case == "plain" else 8)
    B, H, W = 2, 12, 40
    groups = 2 if case == "cat_skip" else 8      # (128 + 64) / 2 = 96 channels per group: the source split is tile aligned
    C0, C1, Cout = (128, 64, 128) if case == "cat_skip" else (128, 0, 128)
    Cmid = 2 * Cout
    a = _r(torch.randn(B, C0, H, W, generator=g)).requires_grad_(True)
    b = _r(torch.randn(B, C1, H, W, generator=g)).requires_grad_(True) if C1 else None
    w0 = torch.randn(Cmid, (C0 + C1) // groups, 3, 3, generator=g, requires_grad=True)
    w1 = torch.randn(Cout, Cmid // groups, 3, 3, generator=g, requires_grad=True)
    ws = torch.randn(Cout, C0 + C1, 1, 1, generator=g, requires_grad=True) if C1 else None
    c = (torch.rand(B, Cmid, generator=g) + 0.5).requires_grad_(True)
    dout = _r(torch.randn(B, Cout, H, W, generator=g))
    s0, s1 = O.cat_mp_weights(C0, C1, 0.5) if C1 else (1.0, 1.0)
    # ---- reference: fp32 autograd on the same (bf16-representable) inputs
    x = torch.cat([s0 * a, s1 * b], 1) if C1 else a
    y0 = O.conv_mp(O.silu_mp(x), w0, groups=groups, training=True)
    y1 = O.conv_mp(O.silu_mp(y0 * c[:, :, None, None]), w1, groups=groups, training=True)
    sk = O.conv_mp(x, ws, training=True) if C1 else x
    out = O.sum_mp(sk, y1, 0.3).clamp(-256, 256)
    leaves = [a, w0, w1, c] + ([b, ws] if C1 else [])
    grads = torch.autograd.grad(out, leaves, dout)
    ref = dict(zip(["dsrc0", "dw_res0", "dw_res1", "dc"] + (["dsrc1", "dw_skip"] if C1 else []), grads))
    # ---- HIP (bf16 activations, fp32 master weights)
    dt = torch.bfloat16
    o, tape = block_forward_train(to_nhwc(a.detach(), dt), to_nhwc(b.detach(), dt) if C1 else None, s0, s1, c.detach().cuda(),
                                  w0.detach().cuda(), w1.detach().cuda(), ws.detach().cuda() if C1 else None, groups, 0.3, 256.0)
    e_fwd = rel_l2(to_nchw(o), out)
    got = block_backward(tape, to_nhwc(dout, dt))
    torch.cuda.synchronize()
    errs = {"fwd": e_fwd}
    for k, r in ref.items():
        v = got[k]
        errs[k] = rel_l2(to_nchw(v) if v.dim() == 4 and k.startswith("dsrc") else v, r)
    print(f"block {case}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # bf16 storage of y0 / a1 / the gradients between the kernels: a few 1e-3 per hop
    assert all(v < 2e-2 for v in errs.values()), errs
