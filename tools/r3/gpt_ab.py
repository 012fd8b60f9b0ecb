#!/usr/bin/env python
"""GPT leg of IndexTTS (prompt pass + greedy decode) with the library named by MI355TTS_LIB: time per token, and the
tokens / hidden rows saved for a bitwise A/B between two builds.

    MI355TTS_LIB=.../libmi355tts_base.so python tools/r3/gpt_ab.py out_base.npz
    python tools/r3/gpt_ab.py out_new.npz ; python tools/r3/gpt_ab.py --cmp out_base.npz out_new.npz
"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "text-to-speech-tts-onnx_amd"))

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        d = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()) if a[k].dtype.kind == "f" else int((a[k] != b[k]).sum())
        print(f"{k}: identical={same} maxdiff/mismatch={d}")
    sys.exit(0)

import torch
from mi355tts.config import IndexGPTConfig
from mi355tts import weights as W
from mi355tts.indextts import IndexGPT

dtype = os.environ.get("AB_DTYPE", "f16")
n_tok = int(os.environ.get("AB_TOKENS", "256"))
gcfg = IndexGPTConfig()
graw = W.synth_state(W.gpt_spec(gcfg), 9527, fast=True)
gpt = IndexGPT(gcfg, graw, dtype=dtype, device=0)
dev = torch.device("cuda:0")
text = (np.arange(30, dtype=np.int32) * 37) % (gcfg.text_tokens - 2) + 2
text_h = gpt.text_embed(text)
mel_h, _ = gpt.mel_embed(gcfg.start_mel_token, 0)
lat = W.synth_normal_fast(3, "lat", (1, 32, gcfg.hidden), std=0.5).astype(np.float32)
pr, cl = gpt.concat(lat, text_h, mel_h)
prompt = torch.from_numpy(pr[0]).to(dev)
toks = torch.zeros((n_tok,), dtype=torch.int32, device=dev)
hid = torch.zeros((n_tok, gcfg.hidden), dtype=torch.float32, device=dev)
for _ in range(2):
    n = gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    n = gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
# prompt pass alone (1 token)
tp = []
t1 = torch.zeros((1,), dtype=torch.int32, device=dev); h1 = torch.zeros((1, gcfg.hidden), dtype=torch.float32, device=dev)
for _ in range(5):
    t0 = time.perf_counter()
    gpt.generate_torch(prompt, 1, t1, h1, stop_tokens=[])
    torch.cuda.synchronize()
    tp.append(time.perf_counter() - t0)
n = gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
torch.cuda.synchronize()
best, bp = min(ts), min(tp)
print(f"lib={os.environ.get('MI355TTS_LIB', 'product')} dtype={dtype} prompt_rows={int(cl[0])} tokens={n}: leg {best*1e3:.2f} ms, "
      f"prompt pass {bp*1e3:.2f} ms, decode {(best-bp)/(n_tok-1)*1e3:.4f} ms/token")
np.savez(sys.argv[1], toks=toks.cpu().numpy(), hid=hid.cpu().numpy())
