import sys; sys.path.insert(0, ".")
import torch
from tests.util import load_golden, rel_l2
from oracle import mel_oracle as M
from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig
t, m = load_golden("mel_stft")
mel = t["mel"][:1]
win = M.hann_power_window(6400, 32.0); fb = M.mel_filterbank(3201, 256, 20.0, 16000.0, 32000)
fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
for n in (1, 2, 3, 4):
    ref = M.mel_to_raw(mel, window=win, hop=256, filters=fb, n_iter=n)
    got = fmt.sample_to_raw(mel, n_fgla_iters=n, quiet=True)
    print("n_iter", n, "gpu vs oracle", rel_l2(got, ref))
# pieces: un-mel
amp = (mel / 2.247 + 1.295).clip(min=0) ** 4
spec = M.unmel(amp, fb)      # (1,2,3201,T)
from dualdiffusion_amd import ops
from dualdiffusion_amd._lib import lib, ptr, check, current_stream
x = mel.cuda().contiguous()
B, Cn, n_mel, T = x.shape
a2 = torch.empty(B * Cn, 1, T, n_mel, device="cuda")
check(lib().ddx_mel_to_amplitude(ptr(x), ptr(a2), B * Cn, n_mel, T, 2.247, 1.295, 4.0, current_stream()))
print("amp", rel_l2(a2[:, 0].permute(0, 2, 1).reshape(1, 2, 256, T), amp))
mags = ops.conv2d(a2, fmt._unmel_weights())
mg = torch.relu(mags[:, 0, :, :3201]).permute(0, 2, 1).reshape(1, 2, 3201, T)
print("unmel", rel_l2(mg, spec))
