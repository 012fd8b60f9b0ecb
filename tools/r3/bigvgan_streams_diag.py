#!/usr/bin/env python
"""Is the side-stream BigVGAN forward run-to-run identical, and how far is it from the one-stream forward?"""
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
B, F = int(os.environ.get("B", 8)), int(os.environ.get("F", 512))
voc = BigVGANVocoder(cfg, blob_device=blob, dtype=dtype, device=0)
mel = torch.from_numpy(W.bigvgan_synthetic_mel(cfg, B, F, 0)).to(dev)
out = torch.empty((B, 1, voc.out_len(F)), dtype=torch.int16, device=dev)
def run(ns):
    _lib.set_option("bigvgan_streams", ns)
    voc.run_torch(mel, out); torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.int32).copy()
ref = run(1)
print(dtype, B, F, "one stream twice identical:", np.array_equal(ref, run(1)))
for ns in (2, 3):
    outs = [run(ns) for _ in range(4)]
    for i, o in enumerate(outs):
        d = np.abs(o - ref)
        print(f"streams={ns} run {i}: differing samples {int((d > 0).sum())} of {d.size}, max |d| {int(d.max())} LSB, same as run 0: {np.array_equal(o, outs[0])}", flush=True)
voc.close()
