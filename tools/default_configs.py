"""The reference's shipped default model configs (config/models/default/*.json), as keyword dicts for the HIP module classes.

The dataclass defaults of the module classes are NOT these (e.g. DualDiffusionVAE_EDM2Config() is 256 x (1,2,3,4) x 2 layers =
21.1 / 57.5 TFLOP per encode / decode, while vae.json is 96 x (1,2,3,5) x 3 = 4.48 / 10.10 TFLOP; UNetConfig() has
channel_mult_emb 4 where unet.json says 3): every tool that quotes a "default model" number builds its models from here.
"""
# config/models/default/unet.json (stale keys use_t_ranges / inpainting / label_dim dropped); same dict as bench.DEFAULT_UNET
DEFAULT_UNET = dict(in_channels=4, out_channels=4, in_channels_emb=512, dropout=0.0, sigma_max=200.0, sigma_min=0.03,
                    sigma_data=1.0, model_channels=256, logvar_channels=128, channel_mult=[1, 2, 3, 4, 5], channel_mult_noise=1,
                    channel_mult_emb=3, channels_per_head=64, num_layers_per_block=2, label_balance=0.5, concat_balance=0.5,
                    res_balance=0.3, attn_balance=0.3, attn_levels=[3, 4], mlp_multiplier=2, mlp_groups=8)
# config/models/default/vae.json (last_global_step dropped)
DEFAULT_VAE = dict(in_channels=2, out_channels=2, latent_channels=4, label_dim=1612, dropout=0.0, target_snr=31.984371183438952,
                   model_channels=96, channel_mult=[1, 2, 3, 5], channel_mult_emb=None, channels_per_head=64, num_layers_per_block=3,
                   res_balance=0.3, attn_balance=0.3, mlp_multiplier=1, mlp_groups=1, add_mid_block_attention=False)
# FLOPs per sample of the default VAE on a 45 s mel spectrogram (2, 256, 5504), SURVEY.md 8d (FlopCounterMode on the reference)
VAE_ENCODE_TFLOP, VAE_DECODE_TFLOP = 4.48, 10.10
