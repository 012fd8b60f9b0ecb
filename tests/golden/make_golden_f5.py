"""F5 golden fixtures: runs the reference's DiT / TextEmbedding / Vocos / STFT_Process module code and
the wrapper classes F5Preprocess / F5Transformer / F5Decode (exec'ed from F5_TTS/Export_F5.py:98-203
where they lie, plus the fold code :321-333 and :390-402) on a reduced-size model with the
synthetic seeded weights.  Output: tests/golden/f5_small.npz (data only)."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, HERE)

from mi355tts.config import F5Config          # noqa: E402
from mi355tts import weights as W             # noqa: E402
import _ref_import as R                       # noqa: E402

SEED = 9527


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def synth_audio(n, seed=SEED):
    tt = np.arange(n) / 24000.0
    a = 0.1 * 32767 * np.sin(2 * np.pi * 220 * tt) + W.synth_normal(seed, "audio", (n,), std=500.0)
    return np.clip(np.round(a), -32768, 32767).astype(np.int16)


def build_ref_f5(cfg: F5Config, state, fp16: bool = False):
    import math
    modules, dit, vmodels, vheads, stft = R.load_f5_ref(fp16)
    model = dit.DiT(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, dim_head=cfg.dim_head, ff_mult=cfg.ff_mult,
                    mel_dim=cfg.mel_dim, text_num_embeds=cfg.text_num_embeds, text_dim=cfg.text_dim,
                    conv_layers=cfg.conv_layers).eval()
    # ConvPositionEmbedding's group count is a constructor default (16); the reduced model uses fewer
    if cfg.pos_conv_groups != 16:
        model.input_embed.conv_pos_embed = modules.ConvPositionEmbedding(dim=cfg.dim, groups=cfg.pos_conv_groups)
    sd = {k[len("transformer."):]: t(v) for k, v in state.items() if k.startswith("transformer.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    f5_model = types.SimpleNamespace(transformer=model)

    backbone = vmodels.VocosBackbone(input_channels=cfg.mel_dim, dim=cfg.vocos_dim,
                                     intermediate_dim=cfg.vocos_intermediate, num_layers=cfg.vocos_layers).eval()
    head = vheads.ISTFTHead(dim=cfg.vocos_dim, n_fft=cfg.n_fft, hop_length=cfg.hop_length, padding="center").eval()

    class Vocos(nn.Module):            # Vocos.decode (vocos/pretrained.py:99-114): backbone -> head
        def __init__(self):
            super().__init__()
            self.backbone, self.head = backbone, head

        def decode(self, f):
            return self.head(self.backbone(f))
    vocos = Vocos()
    vsd = {k[len("vocos."):]: t(v) for k, v in state.items() if k.startswith("vocos.")}
    missing, unexpected = vocos.load_state_dict(vsd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)

    ns = {"torch": torch, "math": math, "torchaudio": sys.modules["torchaudio"], "MAX_SIGNAL_LENGTH": cfg.max_signal_length}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 98, 203, ns,
                 replace=[("self.time_mlp_dim = 1024", f"self.time_mlp_dim = {cfg.dim}")])
    return modules, model, f5_model, vocos, stft, ns


def gen_f5():
    import math
    torch.set_grad_enabled(False)
    from transformers.audio_utils import mel_filter_bank        # before the torchaudio stub is installed
    cfg = F5Config.small()
    state = W.synth_state(W.f5_spec(cfg), SEED)
    modules, model, f5_model, vocos, stft, ns = build_ref_f5(cfg, state)
    out = {}

    # ---- G1: STFT-B / ISTFT-A ---------------------------------------------------------------------
    custom_stft = stft.STFT_Process(model_type="stft_B", n_fft=cfg.n_fft, win_length=cfg.n_fft, hop_len=cfg.hop_length,
                                    max_frames=0, window_type="hann").eval()
    custom_istft = stft.STFT_Process(model_type="istft_A", n_fft=cfg.n_fft, win_length=cfg.n_fft,
                                     hop_len=cfg.hop_length, max_frames=cfg.max_signal_length, window_type="hann").eval()
    xa = W.synth_normal(SEED, "g1.x", (4096,), std=0.3)
    re, im = custom_stft(t(xa).view(1, 1, -1), "reflect")
    out["stft_x"], out["stft_re"], out["stft_im"] = xa, re[0].numpy(), im[0].numpy()
    mag = np.abs(W.synth_normal(SEED, "g1.mag", (cfg.n_freq, 9), std=1.0))
    ph = W.synth_normal(SEED, "g1.ph", (cfg.n_freq, 9), std=2.0)
    out["istft_mag"], out["istft_ph"] = mag, ph
    out["istft_y"] = custom_istft(t(mag)[None], t(ph)[None])[0, 0].numpy()
    out["istft_basis_rows"] = custom_istft.inverse_basis[[0, 1, 7, 512, 513, 514, 700, 1025], 0, :].numpy()
    out["wsi_head"] = custom_istft.window_sum_inv[:2048].numpy()

    # ---- G2/G3: F5Preprocess -----------------------------------------------------------------------
    L, Ttxt = 8192, 14
    audio = synth_audio(L)
    text_ids = (np.abs(W.synth_normal(SEED, "g2.ids", (Ttxt,), std=15.0)).astype(np.int32) % cfg.text_num_embeds)
    text_ids[3] = -1                                       # the pad value of list_str_to_idx -> filler row
    R_len = L // cfg.hop_length + 1
    N = R_len + 15
    pre = ns["F5Preprocess"](f5_model, custom_stft, nfft=cfg.n_fft, n_mels=cfg.mel_dim, sample_rate=cfg.sample_rate,
                             num_head=cfg.heads, head_dim=cfg.dim_head, target_rms=0.15, use_fp16=False)
    torch.manual_seed(0)
    o = pre(t(audio).view(1, 1, -1), t(text_ids).view(1, -1), torch.tensor([N], dtype=torch.long))
    noise, cq, sq, ck, sk, cmt, cmtd, rsl = o
    assert int(rsl) == R_len
    out["pre_audio"], out["pre_text_ids"], out["pre_N"] = audio, text_ids, np.int64(N)
    out["pre_rope_cos_q"], out["pre_rope_sin_q"] = cq[0, 0].numpy(), sq[0, 0].numpy()          # (N, 64)
    assert torch.equal(ck[1, 1], cq[0, 0].T) and torch.equal(sk[0, 1], sq[1, 0].T)
    out["pre_cat_mel_text"], out["pre_cat_mel_text_drop"] = cmt[0].numpy(), cmtd[0].numpy()
    out["pre_ref_signal_len"] = np.int64(int(rsl))
    # cross-check of the restated (un-vendored) HTK filterbank against an installed independent one
    fb2 = mel_filter_bank(cfg.n_freq, cfg.mel_dim, 0.0, 12000.0, cfg.sample_rate, norm=None, mel_scale="htk")
    fb1 = pre.fbank[0].T.numpy()
    assert np.abs(fb1 - fb2).max() < 1e-4, np.abs(fb1 - fb2).max()
    out["fbank_rows"] = pre.fbank[0, [0, 1, 50, 99]].numpy()

    # ---- G4: DiT pieces ---------------------------------------------------------------------------
    math_sf = math.pow(cfg.dim_head, -0.25)
    ns2 = {"torch": torch, "math": math, "f5_model": f5_model, "HEAD_DIM": cfg.dim_head, "use_fp16_transformer": False}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 321, 333, ns2)          # q/k pre-scale fold
    assert abs(ns2["scale_factor"] - math_sf) < 1e-12
    nfe = cfg.nfe_step
    tr = ns["F5Transformer"](f5_model, cfg=cfg.cfg_strength, steps=nfe, sway_coef=cfg.sway_coef, dtype=torch.float32,
                             fuse_step=1)
    out["time_expand"] = tr.time_expand[0].numpy()
    out["delta_t"] = tr.delta_t.numpy()
    noise0 = W.synth_normal(SEED, "g4.noise", (1, N, cfg.mel_dim))
    x0 = t(noise0)
    # one DiT forward at grid point 2 (block-level taps through forward hooks)
    taps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: taps.__setitem__(k, o.numpy().copy()))
             for k, blk in enumerate(model.transformer_blocks)]
    ie = model.input_embed(x0, cmt).numpy()
    pred = model(x=x0, cond=cmt, cond_drop=cmtd, time=tr.time_expand[:, torch.tensor([2])], rope_cos_q=cq, rope_sin_q=sq,
                 rope_cos_k=ck, rope_sin_k=sk)
    for h in hooks:
        h.remove()
    out["dit_noise"], out["dit_input_embed_c"], out["dit_pred_t2"] = noise0[0], ie[0], pred.numpy()
    for k, v in taps.items():
        out[f"dit_block{k}"] = v

    # ---- G5: the sampling loop (NFE grid -> nfe-1 calls) ----------------------------------------------
    x = x0.clone()
    ts = torch.tensor([0], dtype=torch.int32)
    traj = []
    for i in range(nfe - 1):
        x, ts = tr(x, cq, sq, ck, sk, cmt, cmtd, ts)         # mutates in place, like the reference loop
        traj.append(x[0].numpy().copy())
    assert int(ts) == nfe - 1
    out["loop_step1"], out["loop_final"] = traj[0], traj[-1]

    # ---- G6: F5Decode (Vocos fold + ISTFT -> int16) ---------------------------------------------------
    ns3 = {"torch": torch, "vocos": vocos}
    R.exec_lines(R.REF + "/F5_TTS/Export_F5.py", 390, 402, ns3)           # Vocos norm / gamma folds
    dec = ns["F5Decode"](vocos, custom_istft, target_rms=0.15, use_fp16=False)
    den = t(W.synth_normal(SEED, "g6.den", (1, N, cfg.mel_dim), std=0.7))
    mg, pp = vocos.decode(den[:, R_len:].transpose(1, 2))
    out["dec_in"], out["dec_mag"], out["dec_phase"] = den[0].numpy(), mg[0].numpy(), pp[0].numpy()
    out["dec_float"] = custom_istft(mg, pp)[0, 0].numpy()
    out["dec_i16"] = dec(den.clone(), torch.tensor(R_len, dtype=torch.long))[0, 0].numpy()
    assert out["dec_i16"].shape[0] == (N - R_len - 1) * cfg.hop_length
    # ---- G9: end to end: preprocess -> loop -> decode --------------------------------------------------
    out["e2e_i16"] = dec(t(traj[-1])[None].clone(), torch.tensor(R_len, dtype=torch.long))[0, 0].numpy()

    # ---- G10: list_str_to_idx (F5-TTS-ONNX-Inference.py:140-148 exec'ed where it lies): OOV -> 0, pad -1 ----
    ns4 = {"torch": torch}
    R.exec_lines(R.REF + "/F5_TTS/F5-TTS-ONNX-Inference.py", 140, 148, ns4)
    vocab = W.synth_vocab(2545)
    cases = [list("ab c!"), list("Z"), ["a", "\u00e9", "zhong1", " ", "b"]]
    out["g10_ids"] = ns4["list_str_to_idx"](cases, vocab).numpy()

    np.savez_compressed(os.path.join(HERE, "f5_small.npz"), **out)
    print("f5_small.npz:", {k: np.asarray(v).shape for k, v in out.items()})
    for k in ("dit_pred_t2", "loop_final", "dec_float", "dec_mag"):
        print("  ", k, "std %.3f max %.3f" % (out[k].std(), np.abs(out[k]).max()))
    print("   e2e_i16 rms", np.sqrt((out["e2e_i16"].astype(np.float64) ** 2).mean()), "dec_i16 rms",
          np.sqrt((out["dec_i16"].astype(np.float64) ** 2).mean()))


if __name__ == "__main__":
    gen_f5()
