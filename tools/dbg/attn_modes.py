import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts.config import F5Config
from mi355tts import weights as W, _lib
from mi355tts.f5 import F5Engine
from oracle import f5_np as O
cfg = F5Config(dim=256, depth=1, heads=4, dim_head=64, text_dim=64, text_num_embeds=40, conv_layers=1,
               pos_conv_groups=4, vocos_dim=64, vocos_intermediate=128, vocos_layers=1, nfe_step=4)
raw = W.synth_state(W.f5_spec(cfg), 7)
st = W.fold_f5(cfg, raw)
eng = F5Engine(cfg, raw, dtype="f32")
tables = O.time_tables(cfg, st)
for N in (int(a) for a in sys.argv[1:]):
    noise = W.synth_normal(3, f"n{N}", (N, cfg.mel_dim))
    cmt = W.synth_normal(4, f"c{N}", (N, cfg.mel_dim + cfg.text_dim), std=0.7)
    cmtd = W.synth_normal(5, f"d{N}", (N, cfg.mel_dim + cfg.text_dim), std=0.7)
    cos, sin = O.rope_tables(N, 64)
    ref = O.dit_forward(cfg, st, noise, cmt, cmtd, tables[2][1], cos, sin)
    for x3 in (0, 1, 2, 2, 0):
        for q8 in (1, 0):
            _lib.set_option("attn_f32_x3", x3); _lib.set_option("gemm_x3_qkv8", q8)
            y = eng.dit_eval(noise[None], cmt[None], cmtd[None], 1)
            print(N, "attn_x3", x3, "qkv8", q8, "max err", float(np.abs(y - ref).max()), flush=True)
