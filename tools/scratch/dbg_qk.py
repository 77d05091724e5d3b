import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as T
O = T.O
from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
from dualdiffusion_amd.training.unet_grad import UNetTrainer
over = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64, logvar_channels=32)
cfg = O.unet_cfg(**over)
sd = O.random_unet_state(cfg, seed=3, gain_value=0.6, normalized=False)
g = torch.Generator().manual_seed(17)
B, H, W = 2, 16, 32
x_in = torch.randn(B, 4, H, W, generator=g); sigma = torch.tensor([0.4, 3.0])
emb_in = torch.randn(B, O.unet_topology(cfg)["cemb"], generator=g); dD = torch.randn(B, 4, H, W, generator=g)
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "fourier" not in k and not k.startswith("logvar") and not k.startswith("emb_label")}
sd_ref = dict(sd); sd_ref.update(params)
emb_ref = emb_in.clone().requires_grad_(True)
out_ref = O.unet_forward(sd_ref, cfg, x_in, sigma, emb_ref, training=True)
names = list(params)
gref = dict(zip(names, torch.autograd.grad(out_ref, [params[k] for k in names], dD, allow_unused=True)))
unet = UNet(UNetConfig(**over)).requires_grad_(False); unet.load_state_dict(sd, strict=True)
unet = unet.to(device="cuda", dtype=torch.float32).train(True)
tr = UNetTrainer(unet)
out = tr.forward(x_in, sigma, T._Fmt(), emb_in); grads = tr.backward(dD); torch.cuda.synchronize()
k = "enc.block1_layer0.attn_qk.weight"
a = grads[k].reshape(gref[k].shape).cpu().float().flatten(1); r = gref[k].flatten(1)
print(a.shape, "row norms hip", a.norm(dim=1)[:16], "ref", r.norm(dim=1)[:16])
err = (a - r).norm(dim=1) / r.norm(dim=1)
print("rows bad:", (err > 0.05).sum().item(), "of", err.numel(), (err > 0.05).nonzero().flatten()[:40])
# permutation hypothesis
c = torch.nn.functional.normalize(a, dim=1) @ torch.nn.functional.normalize(r, dim=1).T
print("argmax match:", c.argmax(1)[:32])
bank = tr.bank
print("qk job:", [ (e.name, e.qk_head_dim) for e in bank.entries if "attn_qk" in e.name][:3] if hasattr(bank, "entries") else None)
name = [n for n in bank.dwp if "enc.block1_layer0.attn_qk" in n][0]
print("name", name, "parts", name in bank.dwp_parts, bank.dwp_parts[name].shape if name in bank.dwp_parts else None)
G = (bank.dwp_parts[name].sum(0) if name in bank.dwp_parts else bank.dwp[name]).flatten(1).cpu().double()
Wm = dict(unet.named_parameters())[name.replace(".weight", "") + ".weight" if not name.endswith(".weight") else name].data.flatten(1).cpu().double()
rows, fan = Wm.shape; d = 64
od = torch.arange(rows); head = od // (2 * d); rem = od % (2 * d); s = rem // d; dd = rem % d; os_ = head * 2 * d + dd * 2 + s
def expect(grow, wrow):
    x = Wm[wrow]; gg = G[grow]
    ss = (x * x).sum(1, keepdim=True); su = (gg * x).sum(1, keepdim=True)
    rfan = fan ** -0.5; n = ss.sqrt(); nu = 1e-4 + n * rfan
    k = su * rfan / (nu * n)
    return (rfan / nu) * (gg - x * k)
E1 = torch.zeros_like(Wm); E1[os_] = expect(od, os_)
dwhip = bank.dw[name].reshape(rows, fan).cpu().double()
print("hip vs expected(od->os):", ((dwhip - E1).norm() / E1.norm()).item())
E2 = expect(od, od); print("hip vs no-perm:", ((dwhip - E2).norm() / E2.norm()).item())
E3 = torch.zeros_like(Wm); E3[os_] = expect(os_, os_); print("hip vs g-row-os:", ((dwhip - E3).norm() / E3.norm()).item())
print("ref vs expected:", ((r.double() - E1).norm() / E1.norm()).item(), "slices norms", [bank.dwp_parts[name][i].norm().item() for i in range(bank.dwp_parts[name].shape[0])] if name in bank.dwp_parts else None)
