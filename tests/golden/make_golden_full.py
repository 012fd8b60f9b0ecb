#!/usr/bin/env python
"""Full-size golden fixtures (BASELINE.json shapes) from the REFERENCE's own PyTorch module code.

    python tests/golden/make_golden_full.py [f5] [f5fp16] [bigvgan] [zh]      (build container only; ~3 min of torch-CPU each)

* ``f5_full.npz``      F5Config() (dim 1024, 16 heads, depth 22, N = 1126, NFE grid 32): the reference chain
                       F5Preprocess -> 31 x F5Transformer -> F5Decode (wrappers exec'ed from F5_TTS/Export_F5.py:98-203
                       where they lie, q/k fold :321-333, Vocos fold :390-402) on the bench inputs of
                       ``weights.f5_synthetic_inputs`` (utterance 0).  Holds one DiT evaluation, the final sampler state
                       and the int16 waveform; inputs and weights are regenerable from seeds (splitmix64, platform
                       independent), so only the reference's OUTPUTS are stored.
* ``f5_full_fp16.npz`` the same chain as the reference's fp16-transformer export runs it (``use_fp16_transformer``,
                       Export_F5.py:20,88-89,139-140,198-199,321-326,348-349): F5/fp16/modules.py in place of modules.py
                       (q k scores as an fp16 matmul, ``.float() * 100``, fp32 softmax, ``.half()`` probabilities, :467), the
                       extra x0.1 on the q / k projections, F5Preprocess / F5Decode with use_fp16, the whole F5Transformer
                       ``.half()`` (fp16 weights, activations, residual stream and sampler state) — on torch-CPU fp16.
* ``bigvgan_full.npz`` BigVGANConfig() at mel (1,100,512) (BASELINE configs[0]; item 0 of the configs[1] batch) through
                       the reference generator + the int16 wrapper (BigVGAN/Export_BigVGAN.py:37-49).
* ``zh_prompt.npz``    G1 of SURVEY.md §8c: the reference's STFT_Process (stft_B) + the F5Preprocess mel on the first second of
                       the real prompt IndexTTS/example/zh.wav (samples stored: the GPU box has no /root/reference).
Data only.
"""
from __future__ import annotations

import os
import sys
import time
import wave

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, HERE)

from mi355tts.config import BigVGANConfig, F5Config          # noqa: E402
from mi355tts import weights as W                            # noqa: E402
import _ref_import as R                                      # noqa: E402

SEED = 9527
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def gen_f5_full():
    import math
    from make_golden_f5 import build_ref_f5
    cfg = F5Config()
    t0 = time.time()
    state = W.synth_state(W.f5_spec(cfg), SEED)
    modules, model, f5_model, vocos, stft, ns = build_ref_f5(cfg, state)
    print(f"weights + reference modules: {time.time() - t0:.0f} s")
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, 0)
    custom_stft = stft.STFT_Process(model_type="stft_B", n_fft=cfg.n_fft, win_length=cfg.n_fft, hop_len=cfg.hop_length,
                                    max_frames=0, window_type="hann").eval()
    custom_istft = stft.STFT_Process(model_type="istft_A", n_fft=cfg.n_fft, win_length=cfg.n_fft,
                                     hop_len=cfg.hop_length, max_frames=cfg.max_signal_length, window_type="hann").eval()
    pre = ns["F5Preprocess"](f5_model, custom_stft, nfft=cfg.n_fft, n_mels=cfg.mel_dim, sample_rate=cfg.sample_rate,
                             num_head=cfg.heads, head_dim=cfg.dim_head, target_rms=0.15, use_fp16=False)
    o = pre(t(audio[0]).view(1, 1, -1), t(ids[0]).view(1, -1), torch.tensor([N], dtype=torch.long))
    _, cq, sq, ck, sk, cmt, cmtd, rsl = o
    R_len = int(rsl)
    assert R_len == 563 and N == 1126
    ns2 = {"torch": torch, "math": math, "f5_model": f5_model, "HEAD_DIM": cfg.dim_head, "use_fp16_transformer": False}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 321, 333, ns2)          # q/k pre-scale fold
    tr = ns["F5Transformer"](f5_model, cfg=cfg.cfg_strength, steps=cfg.nfe_step, sway_coef=cfg.sway_coef,
                             dtype=torch.float32, fuse_step=1)
    out = {"N": np.int64(N), "ref_signal_len": np.int64(R_len)}
    out["pre_cat_mel_text"] = cmt[0].numpy()
    x0 = t(noise[:1])
    t1 = time.time()
    pred = model(x=x0, cond=cmt, cond_drop=cmtd, time=tr.time_expand[:, torch.tensor([7])], rope_cos_q=cq, rope_sin_q=sq,
                 rope_cos_k=ck, rope_sin_k=sk)
    print(f"one DiT evaluation: {time.time() - t1:.1f} s")
    out["dit_pred_t7"] = pred.numpy()
    x = x0.clone()
    ts = torch.tensor([0], dtype=torch.int32)
    for i in range(cfg.nfe_step - 1):
        x, ts = tr(x, cq, sq, ck, sk, cmt, cmtd, ts)
        if i == 0:
            out["loop_step1"] = x[0].numpy().copy()
    assert int(ts) == cfg.nfe_step - 1
    out["loop_final"] = x[0].numpy().copy()
    print(f"31 evaluations: {time.time() - t1:.0f} s")
    ns3 = {"torch": torch, "vocos": vocos}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 390, 402, ns3)           # Vocos norm / gamma folds
    dec = ns["F5Decode"](vocos, custom_istft, target_rms=0.15, use_fp16=False)
    out["e2e_i16"] = dec(x.clone(), torch.tensor(R_len, dtype=torch.long))[0, 0].numpy()
    assert out["e2e_i16"].shape[0] == (N - R_len - 1) * cfg.hop_length
    np.savez_compressed(os.path.join(HERE, "f5_full.npz"), **out)
    w = out["e2e_i16"].astype(np.float64)
    print("f5_full.npz:", {k: np.asarray(v).shape for k, v in out.items()})
    print("   pred std %.3f  loop_final std %.3f  e2e rms %.0f max %d" % (out["dit_pred_t7"].std(), out["loop_final"].std(),
                                                                          np.sqrt((w ** 2).mean()), np.abs(w).max()))


def gen_f5_full_fp16():
    """The reference's fp16-transformer export semantics at the BASELINE shapes (same inputs / weights as f5_full.npz)."""
    import math
    from make_golden_f5 import build_ref_f5
    cfg = F5Config()
    state = W.synth_state(W.f5_spec(cfg), SEED)
    modules, model, f5_model, vocos, stft, ns = build_ref_f5(cfg, state, fp16=True)
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, 0)
    custom_stft = stft.STFT_Process(model_type="stft_B", n_fft=cfg.n_fft, win_length=cfg.n_fft, hop_len=cfg.hop_length,
                                    max_frames=0, window_type="hann").eval()
    custom_istft = stft.STFT_Process(model_type="istft_A", n_fft=cfg.n_fft, win_length=cfg.n_fft,
                                     hop_len=cfg.hop_length, max_frames=cfg.max_signal_length, window_type="hann").eval()
    pre = ns["F5Preprocess"](f5_model, custom_stft, nfft=cfg.n_fft, n_mels=cfg.mel_dim, sample_rate=cfg.sample_rate,
                             num_head=cfg.heads, head_dim=cfg.dim_head, target_rms=0.15, use_fp16=True)
    o = pre(t(audio[0]).view(1, 1, -1), t(ids[0]).view(1, -1), torch.tensor([N], dtype=torch.long))
    _, cq, sq, ck, sk, cmt, cmtd, rsl = o
    assert cmt.dtype == torch.float16 and cq.dtype == torch.float16
    R_len = int(rsl)
    ns2 = {"torch": torch, "math": math, "f5_model": f5_model, "HEAD_DIM": cfg.dim_head, "use_fp16_transformer": True}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 321, 333, ns2)          # q/k pre-scale fold incl. the fp16 x0.1
    assert ns2["dtype"] == torch.float16 and abs(ns2["scale_factor"] - 0.1 * cfg.dim_head ** -0.25) < 1e-9
    tr = ns["F5Transformer"](f5_model, cfg=cfg.cfg_strength, steps=cfg.nfe_step, sway_coef=cfg.sway_coef,
                             dtype=torch.float16, fuse_step=1)
    tr = tr.half()                                                        # Export_F5.py:348-349
    out = {"N": np.int64(N), "ref_signal_len": np.int64(R_len)}
    x0 = t(noise[:1]).half()                                              # Export_F5.py:139-140 (noise.half())
    t1 = time.time()
    pred = model(x=x0, cond=cmt, cond_drop=cmtd, time=tr.time_expand[:, torch.tensor([7])], rope_cos_q=cq, rope_sin_q=sq,
                 rope_cos_k=ck, rope_sin_k=sk)
    assert pred.dtype == torch.float16
    print(f"one fp16 DiT evaluation: {time.time() - t1:.1f} s")
    out["dit_pred_t7"] = pred.numpy()
    x = x0.clone()
    ts = torch.tensor([0], dtype=torch.int32)
    for i in range(cfg.nfe_step - 1):
        x, ts = tr(x, cq, sq, ck, sk, cmt, cmtd, ts)
        if i == 0:
            out["loop_step1"] = x[0].numpy().copy()
    assert x.dtype == torch.float16 and int(ts) == cfg.nfe_step - 1
    out["loop_final"] = x[0].numpy().copy()
    print(f"31 fp16 evaluations: {time.time() - t1:.0f} s")
    ns3 = {"torch": torch, "vocos": vocos}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 390, 402, ns3)           # Vocos norm / gamma folds
    dec = ns["F5Decode"](vocos, custom_istft, target_rms=0.15, use_fp16=True)
    out["e2e_i16"] = dec(x.clone(), torch.tensor(R_len, dtype=torch.long))[0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "f5_full_fp16.npz"), **out)
    print("f5_full_fp16.npz:", {k: (np.asarray(v).shape, np.asarray(v).dtype) for k, v in out.items()})
    g = np.load(os.path.join(HERE, "f5_full.npz"))                        # the reference's own fp16-vs-fp32 distance
    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return float(np.sqrt(((a - b) ** 2).mean() / (b ** 2).mean()))
    print("   reference fp16 vs reference fp32: dit_pred_t7 rel rms %.3e  loop_final rel rms %.3e  waveform rms %.3e (of full scale)"
          % (rel(out["dit_pred_t7"], g["dit_pred_t7"]), rel(out["loop_final"], g["loop_final"]),
             float(np.sqrt((((out["e2e_i16"].astype(np.float64) - g["e2e_i16"]) / 32767.0) ** 2).mean()))))


def gen_bigvgan_full():
    from make_golden import build_ref_bigvgan
    cfg = BigVGANConfig()
    state = W.synth_state(W.bigvgan_spec(cfg), SEED)
    bv, model = build_ref_bigvgan(cfg, state)
    ns = {"torch": torch}
    R.exec_lines(R.REF + "/BigVGAN/Export_BigVGAN.py", 37, 49, ns)
    wrap = ns["BIGVGAN"](model, True)
    mel = W.bigvgan_synthetic_mel(cfg, 1, 512, 0)
    t0 = time.time()
    wav = wrap(t(mel)).numpy()
    print(f"BigVGAN (1,100,512): {time.time() - t0:.1f} s")
    assert wav.shape == (1, 1, cfg.out_len(512))
    ones = wrap(torch.ones((1, cfg.num_mels, 64))).numpy()          # the reference's own smoke input, Export_BigVGAN.py:165
    np.savez_compressed(os.path.join(HERE, "bigvgan_full.npz"), wav_i16=wav[0, 0], ones64_i16=ones[0, 0])
    w = wav.astype(np.float64)
    print("bigvgan_full.npz: rms %.0f max %d" % (np.sqrt((w ** 2).mean()), np.abs(w).max()))


def gen_zh():
    """G1: the real prompt (IndexTTS/example/zh.wav, 24 kHz mono int16), first second, through the reference STFT_Process
    and the F5Preprocess mel (Export_F5.py:122-125)."""
    import math
    modules, dit, vmodels, vheads, stft = R.load_f5_ref()
    with wave.open(R.REF + "/IndexTTS/example/zh.wav", "rb") as f:
        assert f.getframerate() == 24000 and f.getnchannels() == 1 and f.getsampwidth() == 2
        total = f.getnframes()
        pcm = np.frombuffer(f.readframes(24000), dtype="<i2").copy()
    cfg = F5Config.small()
    custom_stft = stft.STFT_Process(model_type="stft_B", n_fft=cfg.n_fft, win_length=cfg.n_fft, hop_len=cfg.hop_length,
                                    max_frames=0, window_type="hann").eval()
    a = t(pcm).view(1, 1, -1).float() * (1.0 / 32768.0)
    re, im = custom_stft(a, "reflect")
    fb = R.melscale_fbanks(cfg.n_fft // 2 + 1, 0, 12000, cfg.mel_dim, cfg.sample_rate, None, "htk").transpose(0, 1).unsqueeze(0)
    mel = torch.matmul(fb, torch.sqrt(re * re + im * im)).clamp(min=1e-5).log()
    np.savez_compressed(os.path.join(HERE, "zh_prompt.npz"), pcm=pcm, total_samples=np.int64(total),
                        stft_re=re[0].numpy(), stft_im=im[0].numpy(), logmel=mel[0].numpy())
    print("zh_prompt.npz:", pcm.shape, re.shape, mel.shape, "total samples", total)


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"f5", "bigvgan", "zh"}
    if "zh" in what:
        gen_zh()
    if "bigvgan" in what:
        gen_bigvgan_full()
    if "f5" in what:
        gen_f5_full()
    if "f5fp16" in what:
        gen_f5_full_fp16()
