#!/bin/bash
# lane ^ 32 exchange of the attention softmax: v_permlane32_swap through inline asm (in-tree build) against ds_bpermute (the previous
# build, MI355TTS_LIB): bit-identity of a full-size DiT evaluation between the two libraries, then a same-box A/B of the launch time
python - <<'P'
import os, subprocess, sys, numpy as np
code = '''
import sys, numpy as np
sys.path.insert(0, "text-to-speech-tts-onnx_amd")
from mi355tts.config import F5Config
from mi355tts import weights as W
from mi355tts.f5 import F5Engine
cfg = F5Config()
raw = W.synth_state(W.f5_spec(cfg), 9527)
audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 2, 0)
for dt in ("f32", "bf16", "f16"):
    eng = F5Engine(cfg, raw, dtype=dt)
    o = eng.preprocess(audio[0].reshape(1, 1, -1), ids[0].reshape(1, -1), np.array([N]), noise=noise[0])
    p = eng.dit_eval(noise[:1], o["cat_mel_text"], o["cat_mel_text_drop"], 7)
    np.save(sys.argv[1] + "_" + dt + ".npy", p)
    eng.close()
'''
for tag, lib in (("new", None), ("old", os.path.abspath("text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_shfl.so"))):
    env = dict(os.environ)
    if lib: env["MI355TTS_LIB"] = lib
    subprocess.check_call([sys.executable, "-c", code, "/tmp/pred_" + tag], env=env)
for dt in ("f32", "bf16", "f16"):
    a, b = np.load(f"/tmp/pred_new_{dt}.npy"), np.load(f"/tmp/pred_old_{dt}.npy")
    print(dt, "full-size DiT evaluation, permlane build == shuffle build:", np.array_equal(a, b), float(np.abs(a - b).max()))
P
for lib in new old new old; do
  if [ $lib = old ]; then export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_shfl.so; else unset MI355TTS_LIB; fi
  python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary > /tmp/a.json 2>/dev/null
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary > /tmp/b.json 2>/dev/null
  python - <<P
import json
for f, n in (("/tmp/a.json", "bf16 x 8"), ("/tmp/b.json", "fp32 x 1")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    a=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("attn")][0]
    print("$lib", n, round(d["ms_per_step"],1), "ms; attention", round(a["avg_launch_us"],2), "us")
P
done
