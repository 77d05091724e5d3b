"""One EDM2 block forward + backward on the HIP kernels (dualdiffusion_amd.training.block_grad) against torch autograd
through the oracle's block (oracle.edm2_oracle.block_forward, training mode: forced weight norm inside the forward)."""
import pytest
import torch

from oracle import edm2_oracle as O
from tests.util import rel_l2, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu


def _r(x):
    return x.to(torch.bfloat16).float()


CASES = {
    # name: (flavor, resample, C0, C1, Cout, groups, skip conv)
    "dec_cat_skip": ("dec", "keep", 128, 64, 128, 2, True),    # (128 + 64) / 2 = 96 channels per group: the source split is tile aligned
    "dec_plain": ("dec", "keep", 128, 0, 128, 8, False),
    "dec_up": ("dec", "up", 128, 0, 128, 8, False),
    "dec_up_skip": ("dec", "up", 128, 0, 128, 8, True),        # up block with a skip conv: the skip branch runs at the source size
    "enc_down_skip": ("enc", "down", 64, 0, 128, 8, True),
    "enc_plain": ("enc", "keep", 128, 0, 128, 8, False),
    "dec_attn": ("dec", "keep", 128, 0, 128, 8, False),
    "enc_attn_skip": ("enc", "keep", 64, 0, 128, 8, True),
}


@pytest.mark.parametrize("case", list(CASES))
def test_block_forward_backward(case):
    from dualdiffusion_amd.training.block_grad import BlockWeightsT, block_backward, block_forward_train
    flavor, resample, C0, C1, Cout, groups, has_skip = CASES[case]
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    B, H, W, Cemb = 2, 12, 40, 96
    iH, iW = {"keep": (H, W), "up": (H // 2, W // 2), "down": (H * 2, W * 2)}[resample]
    Cmid = 2 * Cout
    a = _r(torch.randn(B, C0, iH, iW, generator=g)).requires_grad_(True)
    b = _r(torch.randn(B, C1, iH, iW, generator=g)).requires_grad_(True) if C1 else None
    emb = torch.randn(B, Cemb, generator=g).requires_grad_(True)
    Cx = Cout if (flavor == "enc" and has_skip) else C0 + C1      # channels seen by conv_res0
    sd = {"blk.conv_res0.weight": torch.randn(Cmid, Cx // groups, 3, 3, generator=g),
          "blk.conv_res1.weight": torch.randn(Cout, Cmid // groups, 3, 3, generator=g),
          "blk.emb_linear.weight": torch.randn(Cmid, Cemb // groups, 1, 1, generator=g),
          "blk.emb_gain": torch.tensor(0.6)}
    if has_skip:
        sd["blk.conv_skip.weight"] = torch.randn(Cout, C0 + C1, 1, 1, generator=g)
    attn = "attn" in case
    heads = Cout // 64
    if attn:
        sd.update({"blk.attn_qk.weight": torch.randn(2 * Cout, Cout, 1, 1, generator=g), "blk.attn_v.weight": torch.randn(Cout, Cout, 1, 1, generator=g),
                   "blk.attn_proj.weight": torch.randn(Cout, Cout, 1, 1, generator=g),
                   "blk.emb_linear_qk.weight": torch.randn(Cout, Cemb, 1, 1, generator=g), "blk.emb_gain_qk": torch.tensor(0.5),
                   "blk.emb_linear_v.weight": torch.randn(Cout, Cemb, 1, 1, generator=g), "blk.emb_gain_v": torch.tensor(-0.4)})
    for v in sd.values():
        v.requires_grad_(True)
    dout = _r(torch.randn(B, Cout, H, W, generator=g))
    s0, s1 = O.cat_mp_weights(C0, C1, 0.5) if C1 else (1.0, 1.0)
    # ---- reference: fp32 autograd through the oracle's block on the same (bf16-representable) inputs
    x = O.cat_mp(a, b, 0.5) if C1 else a
    if has_skip:
        out = O.block_forward(sd, "blk", x, emb[:, :, None, None], flavor=flavor, resample=resample, attention=attn, heads=heads, groups=groups,
                              training=True)
    else:   # the oracle indexes conv_skip unconditionally for its flavor: blocks without one are the identity there
        xr = O.resample2x(x, resample)
        if flavor == "enc":
            xr = O.rms_normalize(xr, dims=[1])
        y = O.conv_mp(O.silu_mp(xr), sd["blk.conv_res0.weight"], groups=groups, training=True)
        c = O.conv_mp(emb[:, :, None, None], sd["blk.emb_linear.weight"], gain=sd["blk.emb_gain"], groups=groups, training=True) + 1.0
        y = O.conv_mp(O.silu_mp(y * c), sd["blk.conv_res1.weight"], groups=groups, training=True)
        out = O.sum_mp(xr, y, 0.3)
        if attn:   # oracle.block_forward lines 142-150
            e4 = emb[:, :, None, None]
            cq = O.conv_mp(e4, sd["blk.emb_linear_qk.weight"], gain=sd["blk.emb_gain_qk"], training=True) + 1.0
            qk_ = O.conv_mp(out * cq, sd["blk.attn_qk.weight"], training=True)
            v_ = O.conv_mp(out, sd["blk.attn_v.weight"], training=True)
            cv = O.conv_mp(e4, sd["blk.emb_linear_v.weight"], gain=sd["blk.emb_gain_v"], training=True) + 1.0
            y = O.conv_mp(O.silu_mp(O.attention_2d(qk_, v_, heads) * cv), sd["blk.attn_proj.weight"], training=True)
            out = O.sum_mp(out, y, 0.3)
        out = out.clamp(-256, 256)
    names = ["din0", "demb", "dw_conv_res0", "dw_conv_res1", "dw_emb_linear", "demb_gain"] + (["din1"] if C1 else []) + (["dw_conv_skip"] if has_skip else [])
    leaves = [a, emb, sd["blk.conv_res0.weight"], sd["blk.conv_res1.weight"], sd["blk.emb_linear.weight"], sd["blk.emb_gain"]] + \
             ([b] if C1 else []) + ([sd["blk.conv_skip.weight"]] if has_skip else [])
    if attn:
        for nm in ("attn_qk", "attn_v", "attn_proj", "emb_linear_qk", "emb_linear_v"):
            names.append(f"dw_{nm}"); leaves.append(sd[f"blk.{nm}.weight"])
        for nm in ("emb_gain_qk", "emb_gain_v"):
            names.append(f"d{nm}"); leaves.append(sd[f"blk.{nm}"])
    ref = dict(zip(names, torch.autograd.grad(out, leaves, dout)))
    # ---- HIP (bf16 activations, fp32 master weights)
    dt = torch.bfloat16
    wts = BlockWeightsT(conv_res0=sd["blk.conv_res0.weight"].detach().cuda(), conv_res1=sd["blk.conv_res1.weight"].detach().cuda(),
                        emb_linear=sd["blk.emb_linear.weight"].detach().cuda(), emb_gain=sd["blk.emb_gain"].detach().cuda().reshape(1),
                        conv_skip=sd["blk.conv_skip.weight"].detach().cuda() if has_skip else None, groups=groups)
    if attn:
        for nm in ("attn_qk", "attn_v", "attn_proj", "emb_linear_qk", "emb_linear_v"):
            setattr(wts, nm, sd[f"blk.{nm}.weight"].detach().cuda())
        wts.emb_gain_qk, wts.emb_gain_v = sd["blk.emb_gain_qk"].detach().cuda().reshape(1), sd["blk.emb_gain_v"].detach().cuda().reshape(1)
        wts.heads = heads
    emb_d = emb.detach().cuda()
    o, tape = block_forward_train(to_nhwc(a.detach(), dt), to_nhwc(b.detach(), dt) if C1 else None, s0, s1, emb_d, wts, flavor=flavor,
                                  resample=resample, res_t=0.3, clip=256.0)
    demb = torch.zeros_like(emb_d)
    got = block_backward(tape, to_nhwc(dout, dt), demb)
    got["demb"] = demb
    torch.cuda.synchronize()
    errs = {"fwd": rel_l2(to_nchw(o), out)}
    for k, r in ref.items():
        v = got[k]
        if k.startswith("din"):
            v = to_nchw(v)
        errs[k] = rel_l2(v.reshape(r.shape) if not k.startswith("din") else v, r)
    print(f"block {case}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # bf16 storage of y0 / a1 / the gradients between the kernels: a few 1e-3 per hop
    # (emb_gain's gradient is ONE scalar summed over signed per-channel terms: cancellation amplifies the relative error)
    assert all(v < (8e-2 if k.startswith("demb_gain") else 2e-2) for k, v in errs.items()), errs


@pytest.mark.parametrize("shape", [(2, 4, 86, 2, 64), (1, 2, 43, 3, 64)], ids=["T344", "T86"])
def test_attention_backward(shape):
    """Attention backward (batched MFMA GEMMs over materialised P / dS + row softmax + normalize backward) against autograd
    through the oracle's attention_2d; T = 86 exercises the zero-padded (T -> 88) score matrices."""
    from dualdiffusion_amd.training.attention_grad import attention_backward
    B, H, W, heads, d = shape
    Cn = heads * d
    g = torch.Generator().manual_seed(B * 100 + W)
    qk = _r(torch.randn(B, 2 * Cn, H, W, generator=g)).requires_grad_(True)       # oracle layout: (head, d, {q,k})
    v = _r(torch.randn(B, Cn, H, W, generator=g)).requires_grad_(True)
    do = _r(torch.randn(B, Cn, H, W, generator=g))
    o = O.attention_2d(qk, v, heads)
    dqk_ref, dv_ref = torch.autograd.grad(o, (qk, v), do)
    perm = lambda t: t.reshape(B, heads, d, 2, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * Cn, H, W)   # -> (head, {q,k}, d)
    dt = torch.bfloat16
    dqk, dv = attention_backward(to_nhwc(perm(qk.detach()), dt), to_nhwc(v.detach(), dt), to_nhwc(do, dt), heads)
    torch.cuda.synchronize()
    e_qk, e_v = rel_l2(to_nchw(dqk), perm(dqk_ref)), rel_l2(to_nchw(dv), dv_ref)
    print(f"attention backward {shape}: dqk {e_qk:.2e}, dv {e_v:.2e}")
    assert e_qk < 2e-2 and e_v < 2e-2


class _Fmt:
    def __init__(self, fmin=20.0, fmax=16000.0):
        from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
        self.ms_freq_scale = FrequencyScale("mel", fmin, fmax, 32000, 3201, 256)


def test_unet_training_forward_backward():
    """Whole EDM2 UNet (2 levels, attention on level 1, every block flavour) forward + backward on the HIP kernels against
    fp32 autograd through the oracle's unet_forward (training mode), for a random upstream gradient dD."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.unet_grad import UNetTrainer
    # model_channels = 256 as in the default config: every mp_cat source split then falls on a 32-channel tile of a group
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=3, gain_value=0.6, normalized=False)
    g = torch.Generator().manual_seed(17)
    B, H, W = 2, 16, 32
    x_in = torch.randn(B, 4, H, W, generator=g)
    sigma = torch.tensor([0.4, 3.0])
    emb_in = torch.randn(B, O.unet_topology(cfg)["cemb"], generator=g)
    dD = torch.randn(B, 4, H, W, generator=g)
    # ---- reference
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "fourier" not in k and not k.startswith("logvar")
              and not k.startswith("emb_label")}
    sd_ref = dict(sd); sd_ref.update(params)
    emb_ref = emb_in.clone().requires_grad_(True)
    out_ref = O.unet_forward(sd_ref, cfg, x_in, sigma, emb_ref, training=True)
    names = list(params)
    gref = dict(zip(names + ["embeddings"], torch.autograd.grad(out_ref, [params[k] for k in names] + [emb_ref], dD, allow_unused=True)))
    # ---- HIP
    unet = UNet(UNetConfig(**over)).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    tr = UNetTrainer(unet)
    out = tr.forward(x_in, sigma, _Fmt(), emb_in)
    grads = tr.backward(dD)
    torch.cuda.synchronize()
    e_fwd = rel_l2(out, out_ref)
    errs = {}
    for k, r in gref.items():
        if r is None:
            continue
        assert k in grads, f"no gradient for {k}"
        errs[k] = rel_l2(grads[k].reshape(r.shape), r)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(f"unet train: fwd {e_fwd:.2e}; {len(errs)} gradients, worst: " + ", ".join(f"{k} {v:.2e}" for k, v in worst))
    assert e_fwd < 2e-2
    rest = {k: v for k, v in errs.items() if "gain" not in k}
    assert all(v < 3e-2 for v in rest.values()), {k: v for k, v in rest.items() if v >= 3e-2}
    # the 0-d gains: each gradient is ONE scalar summed over signed per-channel terms (cancellation), so they are judged
    # together against the largest gain gradient of the model
    gk = [k for k in errs if "gain" in k]
    gmax = max(abs(float(gref[k])) for k in gk)
    bad = {k: (float(grads[k]), float(gref[k])) for k in gk if abs(float(grads[k]) - float(gref[k])) > 3e-2 * abs(float(gref[k])) + 1e-2 * gmax}
    print("gain gradients (hip, ref): " + ", ".join(f"{k.split('.')[-2]}.{k.split('.')[-1]} {float(grads[k]):+.3e}/{float(gref[k]):+.3e}" for k in gk[:8]))
    assert not bad, bad


def test_unet_train_batch():
    """One training batch (embeddings with conditioning dropout, noised + perturbed input, UNet forward/backward, EDM2 loss with
    learned log-variance) against fp32 autograd through the oracle: loss per sample and the gradient of EVERY parameter."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.unet_grad import UNetTrainer
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=5, gain_value=0.5, normalized=False)
    g = torch.Generator().manual_seed(23)
    B, H, W = 2, 16, 32
    samples = torch.randn(B, 4, H, W, generator=g)
    noise, pert = torch.randn(B, 4, H, W, generator=g), torch.randn(B, 4, H, W, generator=g)
    sigma = torch.tensor([0.7, 5.0])
    clap = torch.randn(B, 64, generator=g)
    mask = torch.tensor([True, False])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "fourier" not in k}
    sd_ref = dict(sd); sd_ref.update(params)
    loss_ref = O.unet_train_loss(sd_ref, cfg, samples, clap, sigma, noise, mask, pert, 1.0)
    names = list(params)
    gref = dict(zip(names, torch.autograd.grad(loss_ref.mean(), [params[k] for k in names])))
    unet = UNet(UNetConfig(**over)).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    loss, grads = UNetTrainer(unet).train_batch(samples, clap, sigma, noise, mask, _Fmt(), pert, 1.0)
    torch.cuda.synchronize()
    e_loss = rel_l2(loss, loss_ref)
    missing = [k for k in names if k not in grads]
    assert not missing, f"parameters without a gradient: {missing}"
    errs = {k: rel_l2(grads[k].reshape(gref[k].shape), gref[k]) for k in names}
    worst = sorted(((k, v) for k, v in errs.items() if "gain" not in k), key=lambda kv: -kv[1])[:5]
    print(f"train batch: loss {loss.tolist()} vs {loss_ref.tolist()} (rel {e_loss:.2e}); {len(names)} gradients, worst: " + ", ".join(f"{k} {v:.2e}" for k, v in worst))
    assert e_loss < 1e-2
    assert all(v < 3e-2 for k, v in errs.items() if "gain" not in k), {k: v for k, v in errs.items() if "gain" not in k and v >= 3e-2}
    gk = [k for k in names if "gain" in k]
    gmax = max(abs(float(gref[k])) for k in gk)
    bad = {k: (float(grads[k]), float(gref[k])) for k in gk if abs(float(grads[k]) - float(gref[k])) > 3e-2 * abs(float(gref[k])) + 1e-2 * gmax}
    assert not bad, bad


def test_unet_train_batch_vs_reference_fixture():
    """Same train batch against the REFERENCE's own loss and gradients (tests/golden/unet_train.safetensors)."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.unet_grad import UNetTrainer
    from tests.util import load_golden
    t, m = load_golden("unet_train")
    cfg = O.unet_cfg(**m["cfg"])
    sd = O.random_unet_state(cfg, seed=m["seed"], gain_value=m["gain_value"], normalized=False)
    unet = UNet(UNetConfig(**m["cfg"])).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    loss, grads = UNetTrainer(unet).train_batch(t["samples"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), _Fmt(*m["freq_range"]), t["pert"],
                                                m["input_perturbation"])
    torch.cuda.synchronize()
    assert rel_l2(loss, t["loss"]) < 1e-2
    errs = {k: rel_l2(grads[k].reshape(t[f"grad.{k}"].shape), t[f"grad.{k}"]) for k in m["grads"]}
    print("train batch vs reference fixture: loss rel %.2e; " % rel_l2(loss, t["loss"]) + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(v < (1e-1 if "gain" in k else 3e-2) for k, v in errs.items()), errs


def test_unet_train_batch_fp32_tight_gradient_parity():
    """The fp32 parity path of the whole backward pass (UNetTrainer(compute_dtype=float32): exact-fp32 MFMA forward / data gradient,
    scalar fp32 weight-gradient and attention-backward kernels) against the REFERENCE's loss and gradients: every parameter gradient
    in the fixture, 0-d gains included, at fp32 tolerance -- a 2 % systematic error in one layer's wgrad cannot hide here (the bf16
    path is judged at 3e-2)."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.unet_grad import UNetTrainer
    from tests.util import load_golden
    t, m = load_golden("unet_train")
    cfg = O.unet_cfg(**m["cfg"])
    sd = O.random_unet_state(cfg, seed=m["seed"], gain_value=m["gain_value"], normalized=False)
    unet = UNet(UNetConfig(**m["cfg"])).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    tr = UNetTrainer(unet, compute_dtype=torch.float32)
    loss, grads = tr.train_batch(t["samples"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), _Fmt(*m["freq_range"]), t["pert"],
                                 m["input_perturbation"])
    torch.cuda.synchronize()
    e_loss = rel_l2(loss, t["loss"])
    errs = {}
    for k in m["grads"]:
        ref = t[f"grad.{k}"].double()
        got = grads[k].reshape(ref.shape).double().cpu()
        errs[k] = float((got - ref).norm() / ref.norm().clamp_min(1e-12))
    print("fp32 train batch vs reference fixture: loss rel %.2e; " % e_loss + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert e_loss < 1e-5
    assert all(v < 1e-4 for v in errs.values()), errs
    # and every parameter (not only the 13 in the fixture) against autograd through the oracle
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "fourier" not in k}
    sd_o = dict(sd); sd_o.update(params)
    loss_o = O.unet_train_loss(sd_o, cfg, t["samples"], t["clap"], t["sigma"], t["noise"], t["mask"].bool(), t["pert"], m["input_perturbation"])
    gref = dict(zip(params, torch.autograd.grad(loss_o.mean(), list(params.values()))))
    worst = max((float((grads[k].reshape(gref[k].shape).double().cpu() - gref[k].double()).norm() / gref[k].double().norm().clamp_min(1e-12)), k)
                for k in gref)
    print(f"fp32 train batch vs oracle autograd, all {len(gref)} parameters: worst {worst[1]} {worst[0]:.2e}")
    assert worst[0] < 2e-4, worst


def test_fused_adamw_matches_torch():
    """Global-norm clipping + AdamW (+EMA) multi-tensor kernels against clip_grad_norm_ + torch.optim.AdamW on the CPU."""
    from dualdiffusion_amd.training.optimizer import FusedAdamW, OptimizerConfig
    g = torch.Generator().manual_seed(4)
    shapes = {"a": (300, 17), "b": (5,), "c": (64, 8, 3, 3), "d": ()}
    p_ref = {k: torch.randn(s, generator=g).requires_grad_(True) for k, s in shapes.items()}
    p_hip = {k: v.detach().clone().cuda() for k, v in p_ref.items()}
    ema_ref = {k: v.detach().clone() for k, v in p_ref.items()}
    ema_hip = {k: v.clone().cuda() for k, v in ema_ref.items()}
    cfg = OptimizerConfig(adam_weight_decay=0.01, loss_scale=250.0, max_grad_norm=1.0, dynamic_max_grad_norm_z=3)
    opt = FusedAdamW(p_hip, cfg, ema_hip, ema_beta=0.9)
    ref = torch.optim.AdamW(list(p_ref.values()), lr=3e-3, betas=(cfg.adam_beta1, cfg.adam_beta2), eps=cfg.adam_epsilon, weight_decay=cfg.adam_weight_decay)
    import math
    logmean = logvar = math.log(cfg.max_grad_norm)
    for step in range(4):
        grads = {k: torch.randn(s, generator=g) * (0.002 if step % 2 else 0.02) for k, s in shapes.items()}
        # reference: grads of loss * loss_scale, dynamic clip threshold, clip, step, EMA lerp
        max_norm = math.exp(logmean) + math.exp(logvar / 2) * cfg.dynamic_max_grad_norm_z
        for k, p in p_ref.items():
            p.grad = grads[k].clone() * cfg.loss_scale
        norm_ref = float(torch.nn.utils.clip_grad_norm_(list(p_ref.values()), max_norm))
        ref.step()
        for k in ema_ref:
            ema_ref[k].lerp_(p_ref[k].detach(), 1 - 0.9)
        gn = max(norm_ref, 1e-8); gv = max((gn - math.exp(logmean)) ** 2, 1e-8)
        logmean = logmean * cfg.grad_norm_mean_ema_beta + (1 - cfg.grad_norm_mean_ema_beta) * math.log(gn)
        logvar = logvar * cfg.grad_norm_std_ema_beta + (1 - cfg.grad_norm_std_ema_beta) * math.log(gv)
        norm = opt.step({k: v.cuda() for k, v in grads.items()}, 3e-3)
        assert abs(norm - norm_ref) < 1e-4 * norm_ref
        assert abs(opt.get_max_grad_norm() - (math.exp(logmean) + math.exp(logvar / 2) * 3)) < 1e-6
    for k in shapes:
        assert rel_l2(p_hip[k], p_ref[k].detach()) < 2e-6, k
        assert rel_l2(ema_hip[k], ema_ref[k]) < 2e-6, k


def test_sharded_adamw_hip_matches_fused():
    """training.sharded.ShardedAdamW on the HIP kernels (world of one: the shard is everything but the < 64-element tails) against
    FusedAdamW on the same gradients: four steps, a classic and a feedback EMA, dynamic clip threshold; the parameters are views of the
    flat buffer afterwards.  (The N > 1 arithmetic is covered by the world-2 gloo test on the torch backend.)"""
    from dualdiffusion_amd.training.optimizer import EMASpec, FusedAdamW, OptimizerConfig
    from dualdiffusion_amd.training.sharded import ShardedAdamW
    g = torch.Generator().manual_seed(4)
    shapes = {"dec.a": (300, 17), "dec.c": (64, 8, 3, 3), "enc.w": (96, 40), "enc.b": (5,), "enc.d": ()}
    order = list(shapes)
    init = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    cfg = OptimizerConfig(adam_weight_decay=0.01, loss_scale=250.0, max_grad_norm=1.0, dynamic_max_grad_norm_z=3)
    # fused reference
    p_f = {k: v.clone().cuda() for k, v in init.items()}
    em_f = [EMASpec(name="a", tensors={k: v.clone().cuda() for k, v in init.items()}, beta=0.99),
            EMASpec(name="b", tensors={k: v.clone().cuda() for k, v in init.items()}, beta=0.9, feedback_beta=0.95)]
    fused = FusedAdamW(p_f, cfg, emas=em_f)
    # sharded: parameters + one flat gradient bucket with views, two segments
    params = {k: torch.nn.Parameter(v.clone().cuda(), requires_grad=False) for k, v in init.items()}
    total = sum(v.numel() for v in init.values())
    flat = torch.zeros(total, device="cuda")
    views, off = {}, 0
    for k in order:
        n = init[k].numel()
        views[k] = flat[off:off + n].view(shapes[k])
        off += n
    early = init["dec.a"].numel() + init["dec.c"].numel()
    em_s = [EMASpec(name="a", tensors={k: v.clone().cuda() for k, v in init.items()}, beta=0.99),
            EMASpec(name="b", tensors={k: v.clone().cuda() for k, v in init.items()}, beta=0.9, feedback_beta=0.95)]
    calls = []
    sh = ShardedAdamW([(k, params[k]) for k in order], flat, views, [(0, early), (early, total - early)], cfg, emas=em_s, normalize=lambda: calls.append(1))
    # segment 0 ends off the 64-element grid (9708 = 151 x 64 + 44): its tail is replicated, and segment 1 -- which starts off the grid -- gets a
    # replicated head up to the next multiple of 64 so that its shard, like every shard, starts 256-byte aligned in the bucket
    assert sh.use_hip and sum(hi - lo for lo, hi in sh.tails[0]) == early % 64
    assert sh.tails[1][0] == (early, (early + 63) // 64 * 64) and all(lo % 64 == 0 for (_s, S, lo, _hi) in sh.shards if S > 0)
    for step in range(4):
        grads = {k: torch.randn(s, generator=g) * (0.002 if step % 2 else 0.02) for k, s in shapes.items()}
        betas = [0.99, 0.9]
        n_f = fused.step({k: v.cuda() for k, v in grads.items()}, 3e-3, ema_betas=betas)
        for k in order:
            views[k].copy_(grads[k])
        sh.reduce_segment(0, async_op=True)
        sh.reduce_segment(1)
        n_s = sh.step(3e-3, ema_betas=betas)
        assert abs(n_s - n_f) < 1e-5 * n_f and abs(sh.get_max_grad_norm() - fused.get_max_grad_norm()) < 1e-6
    assert len(calls) == 4
    for k in order:
        assert rel_l2(params[k].data, p_f[k]) < 2e-6, k
        assert params[k].data.data_ptr() == sh.param_flat[sh.offsets[k][0]:].data_ptr()
        for j in range(2):
            assert rel_l2(em_s[j].tensors[k], em_f[j].tensors[k]) < 2e-6, (k, j)


def test_unet_train_steps_reduce_loss():
    """Six full optimizer steps (train batch -> all-reduce (single rank) -> clip + AdamW -> forced weight norm) on a fixed batch:
    the loss goes down and every MPConv weight row is unit-RMS again after each step."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.optimizer import LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=9, gain_value=0.3)
    unet = UNet(UNetConfig(**over)).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    ts = UNetTrainStep(unet, _Fmt(), OptimizerConfig(), LRScheduleConfig(learning_rate=5e-4, lr_warmup_steps=1, lr_reference_steps=1000),
                       input_perturbation=0.0)
    ts.global_step = 1   # past the (1-step) warmup: lr multiplier 1
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 16, 32
    samples, noise = torch.randn(B, 4, H, W, generator=g), torch.randn(B, 4, H, W, generator=g)
    sigma, clap, mask = torch.tensor([0.5, 2.0]), torch.randn(B, 64, generator=g), torch.tensor([True, True])
    losses = []
    for _ in range(6):
        out = ts.step(samples, clap, sigma, noise, mask)
        losses.append(float(out["loss"].mean()))
        w = unet.dec["block0_layer0"].conv_res0.weight.data
        rms = w.flatten(1).square().mean(dim=1).sqrt()
        assert float((rms - 1).abs().max()) < 2e-3
    print(f"train steps: loss {losses}, grad_norm {out['grad_norm']:.3f}")
    assert losses[-1] < losses[0] and min(losses[3:]) < min(losses[:2])


def test_unet_train_step_hipgraph_matches_eager():
    """UNetTrainStep(use_graph=True): the train batch captured into one hipGraph and replayed with new inputs gives the same
    losses / gradient norms / weights as the eager loop (same kernels in the same order; the float atomics of the
    channel-scale gradients make two runs differ by bf16 rounding flips downstream: losses to 5e-4, norms / weights to 2e-3)."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.optimizer import LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=9, gain_value=0.3)
    g = torch.Generator().manual_seed(33)
    B, H, W = 2, 16, 32
    batches = [(torch.randn(B, 4, H, W, generator=g), torch.randn(B, 64, generator=g), torch.rand(B, generator=g) * 2 + 0.2,
                torch.randn(B, 4, H, W, generator=g), torch.tensor([True, False])) for _ in range(3)]
    res = {}
    for use_graph in (False, True):
        unet = UNet(UNetConfig(**over)).requires_grad_(False)
        unet.load_state_dict(sd, strict=True)
        unet = unet.to(device="cuda", dtype=torch.float32).train(True)
        ts = UNetTrainStep(unet, _Fmt(), OptimizerConfig(), LRScheduleConfig(learning_rate=5e-4, lr_warmup_steps=1, lr_reference_steps=1000),
                           use_graph=use_graph)
        ts.global_step = 1
        outs = []
        for (samples, clap, sigma, noise, mask) in batches:
            o = ts.step(samples, clap, sigma, noise, mask)
            outs.append((o["loss"].clone().cpu(), o["grad_norm"]))
        res[use_graph] = (outs, unet.dec["block0_layer0"].conv_res0.weight.data.clone().cpu())
    for (l0, n0), (l1, n1) in zip(res[False][0], res[True][0]):
        assert rel_l2(l1, l0) < 5e-4 and abs(n1 - n0) <= 2e-3 * abs(n0)   # (two runs differ by bf16 rounding flips: float atomics)
    e = rel_l2(res[True][1], res[False][1])
    print(f"hipGraph vs eager: weights after 3 steps rel-L2 {e:.2e}")
    assert e < 2e-3


def test_unet_train_step_bucketed_exchange_world1_rccl(monkeypatch):
    """The two-bucket gradient exchange (decoder bucket all-reduced asynchronously over RCCL while the encoder is
    back-propagated, tail afterwards) on a world_size-1 `nccl` group: SUM over one rank is the identity, so losses / norms /
    weights must match the plain step (same tolerance as eager-vs-graph: float atomics) -- checks the bucket boundary, the hook
    placement (no decoder gradient written after its bucket left) and the stream ordering against RCCL's stream."""
    import socket
    import torch.distributed as dist
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.optimizer import LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=9, gain_value=0.3)
    g = torch.Generator().manual_seed(35)
    B, H, W = 2, 16, 32
    batches = [(torch.randn(B, 4, H, W, generator=g), torch.randn(B, 64, generator=g), torch.rand(B, generator=g) * 2 + 0.2,
                torch.randn(B, 4, H, W, generator=g), torch.tensor([True, False])) for _ in range(3)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = {}
    try:
        for bucketed in (False, True, "graph", "rs_ag", "sharded", "sharded_graph"):
            if bucketed is True:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
                monkeypatch.setenv("DDX_DDP_BUCKETS", "1")
            if bucketed == "rs_ag":      # the explicit reduce_scatter_tensor + all_gather_into_tensor pair over RCCL (eager loop)
                monkeypatch.setenv("DDX_GRAD_EXCHANGE", "rs_ag")
            unet = UNet(UNetConfig(**over)).requires_grad_(False)
            unet.load_state_dict(sd, strict=True)
            unet = unet.to(device="cuda", dtype=torch.float32).train(True)
            # "sharded": reduce-scatter, the parameter pass on this rank's shard of the flat parameter buffer, all-gather (training.sharded)
            ts = UNetTrainStep(unet, _Fmt(), OptimizerConfig(), LRScheduleConfig(learning_rate=5e-4, lr_warmup_steps=1, lr_reference_steps=1000),
                               use_graph=bucketed in ("graph", "sharded_graph"), grad_exchange="sharded" if str(bucketed).startswith("sharded") else None)
            if str(bucketed).startswith("sharded"):
                assert ts.sharded is not None and unet.dec["block0_layer0"].conv_res0.weight.data.data_ptr() >= ts.sharded.param_flat.data_ptr()
            ts.global_step = 1
            tr = ts.trainer
            assert 0 < tr.early_numel < tr.grad_flat.numel()
            # the early bucket is exactly the decoder's tensors
            n_dec = sum(p.numel() for k, p in unet.named_parameters() if k.startswith("dec.") and p.ndim > 0)
            assert tr.early_numel == n_dec
            outs = []
            for (samples, clap, sigma, noise, mask) in batches:
                o = ts.step(samples, clap, sigma, noise, mask)
                outs.append((o["loss"].clone().cpu(), o["grad_norm"]))
            res[bucketed] = (outs, unet.dec["block0_layer0"].conv_res0.weight.data.clone().cpu(),
                             unet.enc["block0_layer0"].conv_res0.weight.data.clone().cpu())
            if bucketed == "graph":
                res["graph_tail"] = ts._graph_tail is not None
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    # "graph": the train batch captured as two hipGraphs cut at the bucket hook, the early collective between the two replays
    assert res["graph_tail"], "graph mode with an exchange must capture the batch in two halves"
    for variant in (True, "graph", "rs_ag", "sharded", "sharded_graph"):
        for (l0, n0), (l1, n1) in zip(res[False][0], res[variant][0]):
            assert rel_l2(l1, l0) < 5e-4 and abs(n1 - n0) <= 2e-3 * abs(n0)   # (two runs differ by bf16 rounding flips: float atomics)
        e_dec, e_enc = rel_l2(res[variant][1], res[False][1]), rel_l2(res[variant][2], res[False][2])
        print(f"bucketed exchange (world 1, rccl, {variant}) vs plain: weights after 3 steps rel-L2 dec {e_dec:.2e} enc {e_enc:.2e}")
        assert e_dec < 2e-3 and e_enc < 2e-3


def test_run_batch_graph_with_accumulation_reports_every_micro_step_loss():
    """UNetTrainStep.run_batch with use_graph=True and two accumulation micro-steps: the per-micro-step losses are those of the eager
    loop (the captured static loss buffer is overwritten by every replay: each micro-step's value is copied out), same gradient norm."""
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.optimizer import LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.sigma_sampler import SigmaSampler, SigmaSamplerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64,
                logvar_channels=32)
    cfg = O.unet_cfg(**over)
    sd = O.random_unet_state(cfg, seed=4, gain_value=0.3)
    g = torch.Generator().manual_seed(8)
    A, Bd, H, W = 2, 2, 16, 32
    samples, clap = torch.randn(A * Bd, 4, H, W, generator=g), torch.randn(A * Bd, 64, generator=g)
    jitter = torch.tensor([0.3])
    res = {}
    for use_graph in (False, True):
        unet = UNet(UNetConfig(**over)).requires_grad_(False)
        unet.load_state_dict(sd, strict=True)
        unet = unet.to(device="cuda", dtype=torch.float32).train(True)
        ts = UNetTrainStep(unet, _Fmt(), OptimizerConfig(), LRScheduleConfig(learning_rate=1e-4, lr_warmup_steps=1, lr_reference_steps=1000),
                           use_graph=use_graph, gradient_accumulation_steps=A, sigma_sampler=SigmaSampler(SigmaSamplerConfig()),
                           conditioning_dropout=0.0)
        ts.global_step = 1
        gen = torch.Generator(device="cuda").manual_seed(77)
        out = ts.run_batch(samples, clap, generator=gen, sigma_jitter=jitter)
        res[use_graph] = (out["loss"].detach().float().cpu().clone(), float(out["grad_norm"]))
    l_e, l_g = res[False][0], res[True][0]
    assert l_e.numel() == A * Bd and l_g.shape == l_e.shape
    assert not torch.allclose(l_e[:Bd], l_e[Bd:]), "the two micro-steps must have different losses for this test to mean anything"
    assert rel_l2(l_g, l_e) < 5e-4, (l_g, l_e)
    assert abs(res[True][1] - res[False][1]) <= 2e-3 * abs(res[False][1])


def test_edm2_loss_in_a_replayed_graph_zeroes_its_workspace():
    """The loss op clears its per-sample sum-of-squares workspace with a KERNEL (csrc/common.hpp: zero_bytes).  With a captured
    hipMemsetAsync node the second replay of a train-batch graph read stale words from whichever tensor had reused the 32-byte
    block (round 4: garbage loss / logvar gradient on replays >= 1, depending on the allocator's layout).  Here the block the
    workspace is freed into is handed to a tensor the graph fills with 3e30 right after the loss: every replay must still give
    the eager result."""
    from dualdiffusion_amd import ops
    torch.manual_seed(4)
    B = 8
    den, tgt = torch.randn(B, 4, 32, 64, device="cuda"), torch.randn(B, 4, 32, 64, device="cuda")
    sigma, logvar = torch.rand(B, device="cuda") + 0.3, torch.randn(B, device="cuda") * 0.1
    loss_e, dd_e, dlv_e = ops.edm2_loss(den, tgt, sigma, logvar, 0.5)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g, dd_g, dlv_g = ops.edm2_loss(den, tgt, sigma, logvar, 0.5)
        junk = [torch.empty(B, device="cuda") for _ in range(4)]     # the allocator hands the freed workspace block to one of these
        for j in junk:
            j.fill_(3e30)
    for rep in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.allclose(loss_g, loss_e, rtol=1e-5, atol=1e-6), f"replay {rep}: {loss_g.tolist()} vs {loss_e.tolist()}"
        assert torch.allclose(dlv_g, dlv_e, rtol=1e-5, atol=1e-6) and torch.equal(dd_g, dd_e)
