import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/text-to-speech-tts-onnx_amd']
from mi355tts.config import F5Config
from mi355tts import weights as W, _lib
from mi355tts.f5 import F5Engine
from oracle import f5_np as O
cfg = F5Config(dim=256, depth=1, heads=4, dim_head=64, text_dim=64, text_num_embeds=40, conv_layers=1, pos_conv_groups=4, vocos_dim=64, vocos_intermediate=128, vocos_layers=1, nfe_step=4)
raw = W.synth_state(W.f5_spec(cfg), 7); st = W.fold_f5(cfg, raw)
eng = F5Engine(cfg, raw, dtype="f32"); tables = O.time_tables(cfg, st)
for N in (67, 257, 700):
    noise = W.synth_normal(3, f"n{N}", (N, cfg.mel_dim)); cmt = W.synth_normal(4, f"c{N}", (N, cfg.mel_dim + cfg.text_dim), std=0.7); cmtd = W.synth_normal(5, f"d{N}", (N, cfg.mel_dim + cfg.text_dim), std=0.7)
    cos, sin = O.rope_tables(N, 64)
    ref = O.dit_forward(cfg, st, noise, cmt, cmtd, tables[2][1], cos, sin)
    for split in (1,0):
        _lib.set_option("attn_split", split)
        for x3 in (2,1,0):
            _lib.set_option("attn_f32_x3", x3)
            g = eng.dit_eval(noise[None], cmt[None], cmtd[None], 1)
            print(N, split, x3, "nan", int(np.isnan(g).sum()), "maxerr", float(np.nanmax(np.abs(g-ref))), flush=True)
