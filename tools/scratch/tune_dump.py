import os, sys, torch
os.environ["DDX_AUTOTUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dualdiffusion_amd import ops
orig = ops._tune_conv
log = []
def spy(d):
    code = orig(d)
    log.append((d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.groups, d.ksize, d.epilogue, bool(d.out2), d.out_act, d.prologue, d.out_head_norm, code))
    return code
ops._tune_conv = spy
dev = torch.device("cuda")
unet = bench.build_model(dev, torch.bfloat16, seed=0)
unet.compile()
B = 4
x = torch.randn(B, 4, 32, 688, device=dev); sigma = torch.rand(B, device=dev) + 0.5; clap = torch.randn(B, 512)
with torch.no_grad():
    emb = unet.get_embeddings(clap, torch.ones(B, dtype=torch.bool))
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    class Fmt:
        ms_freq_scale = FrequencyScale('mel', 20.0, 16000.0, 32000, 3201, 256)
    out = unet(x, sigma, Fmt(), emb)
torch.cuda.synchronize()
print(len(log), "convs tuned")
for e in log:
    if e[-1]:
        print(e)
