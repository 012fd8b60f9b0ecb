#!/usr/bin/env python
"""A/B of the BigVGAN side streams (mi_set_option("bigvgan_streams", 1 | 2 | 3)): f16, mel (8,100,512), no per-launch events."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd")); sys.path.insert(0, ROOT)
from mi355tts.config import BigVGANConfig
from mi355tts import weights as W, _lib
from mi355tts.bigvgan import BigVGANVocoder
cfg = BigVGANConfig()
dev = torch.device("cuda:0")
blob = torch.from_numpy(W.pack_bigvgan(cfg, W.synth_state(W.bigvgan_spec(cfg), 9527))).to(dev)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B, F = 8, 512
voc = BigVGANVocoder(cfg, blob_device=blob, dtype=dtype, device=0)
mel = torch.from_numpy(W.bigvgan_synthetic_mel(cfg, B, F, 0)).to(dev)
out = torch.empty((B, 1, voc.out_len(F)), dtype=torch.int16, device=dev)
ref = None
for rep in range(2):
    for ns in (1, 2, 3):
        _lib.set_option("bigvgan_streams", ns)
        for _ in range(3):
            voc.run_torch(mel, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            voc.run_torch(mel, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        o = out.cpu().numpy().copy()
        if ref is None:
            ref = o
        print(f"{dtype} bigvgan_streams={ns}: {ms:.2f} ms per forward, identical to the one-stream waveform: {np.array_equal(o, ref)}", flush=True)
voc.close()
