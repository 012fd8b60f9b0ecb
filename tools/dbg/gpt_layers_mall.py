#!/usr/bin/env python
"""Is the IndexTTS decode step waiting for HBM at its kernel boundaries?  Decode ms / token / layer for GPT depths whose fp16 weights
do (4, 6 layers: 157 / 236 MB) and do not (12, 24 layers) fit the 256 MB Infinity Cache: if the per-layer time is the same, warming
the cache ahead of a layer (a prefetch stream) cannot help."""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
import numpy as np, torch
from mi355tts.config import IndexGPTConfig
from mi355tts import weights as W
from mi355tts.indextts import IndexGPT
dev = torch.device("cuda", 0)
for layers in (4, 6, 12, 24):
    cfg = dataclasses.replace(IndexGPTConfig(), layers=layers)
    gpt = IndexGPT(cfg, blob=W.pack_gpt(cfg, W.synth_state(W.gpt_spec(cfg), 9527, fast=True)), dtype="f16", device=0)
    P, n_tok = 64, 256
    prompt = torch.from_numpy(W.synth_normal_fast(3, "p", (P, cfg.hidden), std=0.5)).to(dev)
    toks = torch.zeros((n_tok,), dtype=torch.int32, device=dev)
    hid = torch.zeros((n_tok, cfg.hidden), dtype=torch.float32, device=dev)
    for _ in range(2):
        gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    wb = layers * 12 * cfg.hidden ** 2 * 2 / 1e6
    print(f"layers {layers:2d}: weights {wb:6.0f} MB  {dt / n_tok * 1e3:.4f} ms/token  {dt / n_tok / layers * 1e6:.2f} us/token/layer", flush=True)
    gpt.close()
