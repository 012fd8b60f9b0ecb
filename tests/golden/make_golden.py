#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE's own module code
(/root/reference, imported with stubs — see _ref_import.py) on seeded inputs and the synthetic
seeded weights of mi355tts.weights.  Run in the build container only:

    python tests/golden/make_golden.py [bigvgan] [f5]

Fixtures hold data only: inputs, (regenerable) seeds and the reference's outputs.
"""
from __future__ import annotations

import os
import sys

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


# ---------------------------------------------------------------------------------------------
# BigVGAN
# ---------------------------------------------------------------------------------------------
def build_ref_bigvgan(cfg: BigVGANConfig, state):
    bv = R.load_bigvgan_ref()
    model = bv.BigVGAN(R.bigvgan_hparams(cfg), use_cuda_kernel=False)
    model.remove_weight_norm()
    model = model.eval().float()
    sd = {k: t(v) for k, v in state.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("filter" in m) for m in missing), missing   # only the registered FIR buffers
    return bv, model


def gen_bigvgan():
    cfg = BigVGANConfig.small()
    state = W.synth_state(W.bigvgan_spec(cfg), SEED)
    bv, model = build_ref_bigvgan(cfg, state)
    out = {}

    # G7: filter taps + Activation1d (block variant and pad-15 post variant) on (1,8,40)
    act = model.resblocks[3].activations[0]                   # stage 1 block: C = 8
    out["taps_up"] = act.upsample.filter.reshape(-1).numpy()
    out["taps_down"] = act.downsample.lowpass.filter.reshape(-1).numpy()
    C = cfg.stage_channels(1)
    x = W.synth_normal(SEED, "g7.x", (2, C, 40), std=1.5)
    i = 1
    y = torch.cat([act(t(x[b:b + 1]), model.x_shape[i], model.up_filter_pad[i], model.up_pad_zeros[i],
                       model.down_filter_pad[i], model.down_pad_zeros_L[i], model.down_pad_zeros_R[i])
                   for b in range(x.shape[0])], 0)     # pad tables are batch-1
    out["act_x"] = x
    out["act_alpha"] = act.act.alpha.numpy()
    out["act_beta"] = act.act.beta.numpy()
    out["act_y"] = y.numpy()
    xp = x[:1]
    yp = model.activation_post(t(xp), model.x_shape[-1], model.up_filter_pad[-1], model.up_pad_zeros[-1],
                               model.down_filter_pad[-1], model.down_pad_zeros_L[-1], model.down_pad_zeros_R[-1])
    out["post_x"] = xp
    out["post_alpha"] = model.activation_post.act.alpha.numpy()
    out["post_beta"] = model.activation_post.act.beta.numpy()
    out["post_y"] = yp.numpy()
    assert yp.shape[-1] == 40 + 30

    # G8a: the three AMPBlock1 of stage 0 (C=16; k = 3, 7, 11) on (1,16,48)
    C0 = cfg.stage_channels(0)
    xa = W.synth_normal(SEED, "g8.x", (1, C0, 48), std=1.0)
    out["amp_x"] = xa
    for j in range(3):
        ya = model.resblocks[j](t(xa), model.x_shape[0], model.up_filter_pad[0], model.up_pad_zeros[0],
                                model.down_filter_pad[0], model.down_pad_zeros_L[0], model.down_pad_zeros_R[0])
        out[f"amp_y{j}"] = ya.numpy()

    # G8b: whole generator + int16 wrapper (Export_BigVGAN.py:37-49 exec'ed from the reference file)
    ns = {"torch": torch}
    R.exec_lines(R.REF + "/BigVGAN/Export_BigVGAN.py", 37, 49, ns)
    wrap = ns["BIGVGAN"](model, True)
    for name, B, F in (("a", 1, 24), ("b", 2, 7)):
        mel = W.synth_normal(SEED, f"g8.mel.{name}", (B, cfg.num_mels, F), std=1.0)
        # the reference's pad tables are batch-1 (torch.cat with (1,C,pad) zeros), so run per item
        ys, ws = [], []
        for b in range(B):
            ys.append(model(t(mel[b:b + 1])).numpy())
            ws.append(wrap(t(mel[b:b + 1])).numpy())
        out[f"gen_mel_{name}"] = mel
        out[f"gen_y_{name}"] = np.concatenate(ys, 0)
        out[f"gen_i16_{name}"] = np.concatenate(ws, 0)
        assert ys[0].shape[-1] == cfg.out_len(F)
    # the reference's own smoke input (Export_BigVGAN.py:165): np.ones
    mel1 = np.ones((1, cfg.num_mels, 12), dtype=np.float32)
    out["gen_mel_ones"] = mel1
    out["gen_i16_ones"] = wrap(t(mel1)).numpy()
    np.savez_compressed(os.path.join(HERE, "bigvgan_small.npz"), **out)
    print("bigvgan_small.npz:", {k: v.shape for k, v in out.items()})
    i16 = out["gen_i16_a"].astype(np.float64)
    print("  int16 output rms %.1f  max %d" % (np.sqrt((i16 ** 2).mean()), np.abs(i16).max()))


# ---------------------------------------------------------------------------------------------
# IndexTTS graph F (speaker-conditioned vocoder)
# ---------------------------------------------------------------------------------------------
def gen_indextts():
    import torch.nn as nn
    cfg = BigVGANConfig.indextts()
    state = W.synth_state(W.bigvgan_spec(cfg), SEED)
    models = R.load_indextts_bigvgan_ref()
    h = R._AttrDict(gpt_dim=cfg.num_mels, upsample_initial_channel=cfg.upsample_initial_channel,
                    upsample_rates=list(cfg.upsample_rates), upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
                    resblock="1", resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
                    resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes], activation="snakebeta",
                    snake_logscale=True, feat_upsample=False, cond_d_vector_in_each_upsampling_layer=True, num_mels=100,
                    speaker_embedding_dim=512)
    bv = models.BigVGAN(h, use_cuda_kernel=False)
    bv.remove_weight_norm()
    bv = bv.eval().float()
    sd = {k: t(v) for k, v in state.items() if not k.startswith("final_norm.")}
    missing, unexpected = bv.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("filter" in m) or m.startswith("cond") or m.startswith("speaker_encoder") for m in missing), missing

    class GPT(nn.Module):
        def __init__(self):
            super().__init__()
            self.final_norm = nn.LayerNorm(cfg.num_mels)
    gpt = GPT()
    gpt.final_norm.weight.data = t(state["final_norm.weight"])
    gpt.final_norm.bias.data = t(state["final_norm.bias"])
    import types
    idx = types.SimpleNamespace(gpt=gpt, bigvgan=bv)
    ns = {"torch": torch}
    R.exec_lines(R.REF + "/IndexTTS/Export_IndexTTS.py", 292, 314, ns)
    part_F = ns["IndexTTS_F"](idx)
    out = {}
    T_codes = 5
    latent = W.synth_normal(SEED, "ix.latent", (T_codes, cfg.num_mels), std=1.5, mean=0.3)
    conds = [W.synth_normal(SEED, f"ix.cond{i}", (1, cfg.stage_channels(i), 1), std=0.2) for i in range(cfg.num_upsamples)]
    cpre = W.synth_normal(SEED, "ix.cond_pre", (1, cfg.upsample_initial_channel, 1), std=0.2)
    wav = part_F(*[t(c) for c in conds], t(cpre), t(latent))
    out["latent"] = latent
    for i, c in enumerate(conds):
        out[f"cond{i}"] = c
    out["cond_pre"] = cpre
    out["wav_i16"] = wav.numpy()
    assert wav.shape == (1, 1, (T_codes - 2) * cfg.hop + 30), wav.shape
    np.savez_compressed(os.path.join(HERE, "indextts_f.npz"), **out)
    w = out["wav_i16"].astype(np.float64)
    print("indextts_f.npz:", {k: v.shape for k, v in out.items()}, "rms", np.sqrt((w ** 2).mean()), "max", np.abs(w).max())


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"bigvgan", "f5", "indextts"}
    if "indextts" in what:
        gen_indextts()
    if "bigvgan" in what:
        gen_bigvgan()
    if "f5" in what:
        from make_golden_f5 import gen_f5
        gen_f5()
