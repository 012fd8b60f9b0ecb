#!/usr/bin/env python
"""Do two persistent-kernel chains on two HIP streams overlap their launch tails?  Two F5 engines (two handles = two streams, each
replaying its own hipGraph), one utterance each, driven from two host threads (ctypes releases the GIL), against one engine run
twice in a row.  If the per-utterance time of the concurrent pair is well below the single-engine time, splitting ONE evaluation's
CFG branches over two streams is worth building (LOG.md round 4).

    python tools/dbg/two_engines_overlap.py [dtype] [steps] [U per engine] [engines]
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "text-to-speech-tts-onnx_amd")]
import numpy as np, torch
from mi355tts.config import F5Config
from mi355tts import weights as W
from mi355tts.f5 import F5Engine

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
U = int(sys.argv[3]) if len(sys.argv) > 3 else 1
NE = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L = int(sys.argv[5]) if len(sys.argv) > 5 else 144000
cfg = F5Config()
blob = torch.from_numpy(W.pack_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527, fast=True))).cuda()
audio, ids, N, noise = W.f5_synthetic_inputs(cfg, U, 0, L=L)
dev = torch.device("cuda", 0)
ta, ti, tn = torch.from_numpy(audio).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(noise).to(dev)
engs = [F5Engine(cfg, blob_device=blob, dtype=dtype) for _ in range(NE)]
R = cfg.ref_frames(audio.shape[1])
outs = [torch.empty((U, 1, (N - R - 1) * cfg.hop_length), dtype=torch.int16, device=dev) for _ in range(NE)]
for e, o in zip(engs, outs):
    for _ in range(3):
        e.synthesize_torch(ta, ti, N, noise=tn, out=o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    engs[0].synthesize_torch(ta, ti, N, noise=tn, out=outs[0])
torch.cuda.synchronize()
single = (time.perf_counter() - t0) / steps

def worker(i):
    for _ in range(steps):
        engs[i].synthesize_torch(ta, ti, N, noise=tn, out=outs[i])

t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i,)) for i in range(NE)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
pair = (time.perf_counter() - t0) / steps
print(f"{dtype} U={U} N={N}: one engine {single * 1e3:.1f} ms per step ({single * 1e3 / U:.1f} per utterance); {NE} engines on {NE} streams {pair * 1e3:.1f} ms per round "
      f"= {pair * 1e3 / (NE * U):.1f} ms per utterance ({single / U / (pair / (NE * U)):.2f}x)")
assert NE < 2 or torch.equal(outs[0], outs[1])
