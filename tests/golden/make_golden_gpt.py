"""Golden vectors for the IndexTTS acoustic GPT-2 graphs (B, C, D, E).  Build-container only.

Runs the reference wrapper classes IndexTTS_B/C/D/E (exec'd from /root/reference IndexTTS/Export_IndexTTS.py:203-289
where they lie) over a stand-in ``indexTTS.gpt`` whose transformer blocks are real Hugging Face ``GPT2Block`` modules
(the un-vendored upstream model is a ``GPT2Model``), loaded with the seeded synthetic weights of
``mi355tts.weights.gpt_spec``.  Two textual patches make the wrappers runnable outside ``torch.onnx.export`` tracing,
where ``tensor.shape[i]`` is a tensor: ``x.shape[-1].unsqueeze(0)`` -> ``torch.tensor([x.shape[-1]])``.

    python tests/golden/make_golden_gpt.py      # writes tests/golden/indextts_gpt.npz
"""
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
sys.path.insert(0, ROOT)

import _ref_import as R                      # noqa: E402
from mi355tts import weights as W            # noqa: E402
from mi355tts.config import IndexGPTConfig   # noqa: E402

SEED = 9527


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def build_ref_gpt(cfg: IndexGPTConfig, state):
    from transformers import GPT2Config
    from transformers.models.gpt2.modeling_gpt2 import GPT2Block
    hf = GPT2Config(vocab_size=cfg.mel_codes, n_positions=cfg.max_seq, n_embd=cfg.hidden, n_layer=cfg.layers,
                    n_head=cfg.heads, n_inner=cfg.inner, activation_function="gelu_new",
                    layer_norm_epsilon=cfg.ln_eps)
    hf._attn_implementation = "eager"

    class Pos(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.emb = nn.Embedding(n, cfg.hidden)

    class Transformer(nn.Module):
        def __init__(self):
            super().__init__()
            self.h = nn.ModuleList([GPT2Block(hf, layer_idx=i) for i in range(cfg.layers)])
            self.ln_f = nn.LayerNorm(cfg.hidden, eps=cfg.ln_eps)

    class Inference(nn.Module):
        def __init__(self):
            super().__init__()
            self.embeddings = nn.Embedding(cfg.mel_codes, cfg.hidden)
            self.text_pos_embedding = Pos(cfg.max_mel_pos)
            self.transformer = Transformer()
            self.lm_head = nn.Sequential(nn.LayerNorm(cfg.hidden, eps=cfg.ln_eps), nn.Linear(cfg.hidden, cfg.mel_codes))

    class GPT(nn.Module):
        def __init__(self):
            super().__init__()
            self.text_embedding = nn.Embedding(cfg.text_tokens, cfg.hidden)
            self.text_pos_embedding = Pos(cfg.max_text_pos)
            self.inference_model = Inference()

    gpt = GPT().eval().float()
    missing, unexpected = gpt.load_state_dict({k: t(v) for k, v in state.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith(".attn.bias") or m.endswith(".attn.masked_bias") for m in missing), missing
    return gpt


def gen_gpt():
    cfg = IndexGPTConfig.small()
    state = W.synth_state(W.gpt_spec(cfg), SEED)
    out = {}
    patches = (("text_ids.shape[-1].unsqueeze(0)", "torch.tensor([text_ids.shape[-1]])"),
               ("concat_hidden_state.shape[1].unsqueeze(0)", "torch.tensor([concat_hidden_state.shape[1]])"))
    ns = {"torch": torch}
    R.exec_lines(R.REF + "/IndexTTS/Export_IndexTTS.py", 203, 289, ns, replace=patches)

    with torch.no_grad():
        idx = types.SimpleNamespace(gpt=build_ref_gpt(cfg, state))
        part_B, part_C, part_D = ns["IndexTTS_B"](idx), ns["IndexTTS_C"](idx), ns["IndexTTS_D"]()
        text_ids = np.array([[5, 17, 3, 22, 9, 30]], np.int32)
        text_h = part_B(t(text_ids))
        out["text_ids"] = text_ids
        out["B_text_hidden"] = text_h.numpy()
        hs_c, gen_len = part_C(torch.tensor([[cfg.start_mel_token]], dtype=torch.int32), torch.tensor([0]))
        out["C_hidden_0"] = hs_c.numpy().copy()
        assert int(gen_len) == 1
        conds = W.synth_normal(SEED, "gpt.conds_latent", (1, 4, cfg.hidden), std=0.5)
        out["conds_latent"] = conds
        hs, concat_len = part_D(t(conds), text_h, hs_c)
        out["D_hidden"] = hs.numpy().copy()
        out["D_len"] = concat_len.numpy()

        # E is built last: its constructor rewrites the attention weights of the shared blocks in place
        part_E = ns["IndexTTS_E"](idx, cfg.layers, cfg.max_seq)
        H, D = cfg.heads, cfg.head_dim
        keys = [torch.zeros((H, D, 0))] * cfg.layers
        values = [torch.zeros((H, 0, D))] * cfg.layers
        pen = torch.ones((1, cfg.mel_codes))
        hist = torch.tensor([0])
        ids_len = concat_len.clone()
        flag = torch.tensor([1], dtype=torch.int8)
        toks, hid = [], []
        reset = 0
        n = 0
        rep, prange = 0.7, 3
        while n < 12:
            res = part_E(*keys, *values, hist, pen, ids_len, hs.clone(), flag)
            keys = [k.clone() for k in res[:cfg.layers]]
            values = [v.clone() for v in res[cfg.layers:2 * cfg.layers]]
            kvl, last, tok = res[-3], res[-2], res[-1]
            if n == 0:
                out["E0_last_hidden"] = last.numpy().copy()
                out["E0_key0"] = keys[0].numpy().copy()
                out["E0_value1"] = values[1].numpy().copy()
            tk = int(tok)
            toks.append(tk)
            hid.append(last.numpy().copy())
            n += 1
            flag = torch.tensor([0], dtype=torch.int8)
            ids_len = torch.tensor([1])
            hist = kvl
            pen[:, tk] = rep
            if n > prange and toks[reset] != tk:
                pen[:, toks[reset]] = 1.0
                reset += 1
            hs, gen_len = part_C(tok, gen_len)
        out["gen_tokens"] = np.array(toks, np.int32)
        out["gen_hidden"] = np.concatenate(hid, axis=0)
        out["gen_penalty"] = pen.numpy().copy()
        out["gen_key0"] = keys[0].numpy().copy()
        out["gen_value1"] = values[1].numpy().copy()
        out["gen_params"] = np.array([rep, prange], np.float32)

        # one mid-sequence step with an explicit cache and a non-trivial penalty (single-call parity)
        pen2 = t(W.synth_normal(SEED, "gpt.pen2", (1, cfg.mel_codes), std=0.2, mean=1.0))
        hs2 = t(W.synth_normal(SEED, "gpt.hs2", (1, 1, cfg.hidden), std=0.7))
        res = part_E(*keys, *values, hist, pen2, torch.tensor([1]), hs2.clone(), torch.tensor([0], dtype=torch.int8))
        out["S_pen"] = pen2.numpy()
        out["S_hidden_in"] = hs2.numpy()
        out["S_last_hidden"] = res[-2].numpy().copy()
        out["S_token"] = res[-1].numpy().copy()
        # all keys/values of the cache that step started from (what a caller would hand in)
        out["S_keys_in"] = np.stack([k.numpy() for k in keys])
        out["S_values_in"] = np.stack([v.numpy() for v in values])
    np.savez_compressed(os.path.join(HERE, "indextts_gpt.npz"), **out)
    print("indextts_gpt.npz:", {k: v.shape for k, v in out.items()})
    print("tokens", toks)


if __name__ == "__main__":
    gen_gpt()
