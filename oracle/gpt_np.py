"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the IndexTTS acoustic GPT-2 graphs B, C, D, E and the
autoregressive driver loop around them.

Plain numpy restatement of /root/reference IndexTTS/Export_IndexTTS.py:203-289 (wrapper classes IndexTTS_B..E) and
IndexTTS/Inference_IndexTTS_ONNX.py:716-783 (the decode loop).  The GPT-2 block itself is Hugging Face
``transformers`` ``GPT2Block`` (un-vendored dependency of upstream IndexTTS: ln_1 -> attention -> residual -> ln_2 ->
c_fc -> gelu_new -> c_proj -> residual; LayerNorm eps 1e-5); its published definition is restated here and pinned
by tests/golden/indextts_gpt.npz, which tests/golden/make_golden_gpt.py generates by running the reference wrapper
classes over real ``GPT2Block`` modules.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
State dict keys follow ``indexTTS.gpt`` as the wrappers read it (mi355tts.weights.gpt_spec); Conv1D weights are
(in, out) like upstream.
"""
from __future__ import annotations

import math

import numpy as np


def layer_norm(x, w, b, eps=1e-5):
    """torch.nn.LayerNorm over the last axis (biased variance)."""
    x = np.asarray(x, np.float32)
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps)) * w + b).astype(np.float32)


def gelu_new(x):
    """transformers.activations.NewGELUActivation (GPT2Config.activation_function default 'gelu_new')."""
    x = np.asarray(x, np.float32)
    return (0.5 * x * (1.0 + np.tanh(np.float32(math.sqrt(2.0 / math.pi)) * (x + np.float32(0.044715) * x ** 3)))
            ).astype(np.float32)


def softmax(x):
    x = x - x.max(-1, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def fold_layer(cfg, st, i):
    """IndexTTS_E.__init__ per-layer weight surgery (Export_IndexTTS.py:252-268): returns per-head q/k/v weights
    (H, hidden, D), biases (H, 1, D), c_proj (H, D, hidden) and its bias."""
    h, H, D = cfg.hidden, cfg.heads, cfg.head_dim
    p = f"inference_model.transformer.h.{i}."
    sc = np.float32(float(D) ** -0.25)
    w = np.array(st[p + "attn.c_attn.weight"], np.float32).T.copy()      # (3h, h)
    b = np.array(st[p + "attn.c_attn.bias"], np.float32).copy()
    w[: 2 * h] *= sc
    b[: 2 * h] *= sc
    out = {}
    for j, nm in enumerate("qkv"):
        out["w" + nm] = np.ascontiguousarray(w[j * h:(j + 1) * h].reshape(H, D, h).transpose(0, 2, 1))
        out["b" + nm] = b[j * h:(j + 1) * h].reshape(H, 1, D)
    wp = np.array(st[p + "attn.c_proj.weight"], np.float32).T            # (out=h, in=h)
    out["wo"] = np.ascontiguousarray(wp.reshape(h, H, D).transpose(1, 2, 0))
    out["bo"] = np.array(st[p + "attn.c_proj.bias"], np.float32).reshape(1, 1, -1)
    return out


def graph_b(cfg, st, text_ids):
    """IndexTTS_B.forward (Export_IndexTTS.py:210-214): text_ids (1, n) int32 -> (1, n + 2, hidden)."""
    ids = np.concatenate([[0], np.asarray(text_ids).reshape(-1), [1]]).astype(np.int64)
    emb = st["text_embedding.weight"][ids] + st["text_pos_embedding.emb.weight"][: len(ids)]
    return emb[None].astype(np.float32)


def graph_c(cfg, st, gpt_ids, gen_len):
    """IndexTTS_C.forward (:222-225): gpt_ids (1, 1) int32, gen_len (1,) -> ((1, 1, hidden), gen_len + 1)."""
    g = int(np.asarray(gen_len).reshape(-1)[0])
    hs = st["inference_model.embeddings.weight"][int(np.asarray(gpt_ids).reshape(-1)[0])] \
        + st["inference_model.text_pos_embedding.emb.weight"][g]
    return hs.reshape(1, 1, -1).astype(np.float32), np.array([g + 1], np.int64)


def graph_d(embed_x, embed_y, embed_z):
    """IndexTTS_D.forward (:233-235)."""
    c = np.concatenate([embed_x, embed_y, embed_z], axis=1).astype(np.float32)
    return c, np.array([c.shape[1]], np.int64)


def graph_e(cfg, st, keys, values, history_len, repeat_penality, ids_len, hidden_state, attention_mask, folds=None):
    """IndexTTS_E.forward (:270-289).

    keys[i] (H, D, hist), values[i] (H, hist, D); hidden_state (1, ids_len, hidden); repeat_penality (1, codes);
    attention_mask 0/1.  Returns (out_keys, out_values, kv_seq_len, last_hidden_state (1, hidden), max_logit_id
    (1, 1) int32)."""
    hs = np.array(hidden_state, np.float32)
    ids_len = int(ids_len)
    kv = int(history_len) + ids_len
    # (1 - tril) * -128, sliced [:ids_len, :kv]: note the row index is the position inside the new block (:247, :274)
    i_idx = np.arange(ids_len)[:, None]
    j_idx = np.arange(kv)[None, :]
    mask = np.where(j_idx > i_idx, np.float32(-128.0), np.float32(0.0)) * np.float32(int(attention_mask))
    out_k, out_v = [], []
    for i in range(cfg.layers):
        p = f"inference_model.transformer.h.{i}."
        f = folds[i] if folds is not None else fold_layer(cfg, st, i)     # (the export does this once, at build time)
        xn = layer_norm(hs, st[p + "ln_1.weight"], st[p + "ln_1.bias"], cfg.ln_eps)          # (1, ids, h)
        q = np.matmul(xn, f["wq"]) + f["bq"]                                                    # (H, ids, D)
        k = (np.matmul(xn, f["wk"]) + f["bk"]).transpose(0, 2, 1)                               # (H, D, ids)
        v = np.matmul(xn, f["wv"]) + f["bv"]
        k = np.concatenate([np.asarray(keys[i], np.float32), k], axis=2)
        v = np.concatenate([np.asarray(values[i], np.float32), v], axis=1)
        out_k.append(k)
        out_v.append(v)
        a = np.matmul(softmax(np.matmul(q, k) + mask), v)                                       # (H, ids, D)
        a = np.matmul(a, f["wo"]).sum(axis=0, keepdims=True) + f["bo"]
        hs = hs + a
        m = layer_norm(hs, st[p + "ln_2.weight"], st[p + "ln_2.bias"], cfg.ln_eps)
        m = gelu_new(m @ np.asarray(st[p + "mlp.c_fc.weight"], np.float32) + st[p + "mlp.c_fc.bias"])
        hs = hs + (m @ np.asarray(st[p + "mlp.c_proj.weight"], np.float32) + st[p + "mlp.c_proj.bias"])
    last = layer_norm(hs[:, -1], st["inference_model.transformer.ln_f.weight"],
                      st["inference_model.transformer.ln_f.bias"], cfg.ln_eps)                  # (1, h)
    z = layer_norm(last, st["inference_model.lm_head.0.weight"], st["inference_model.lm_head.0.bias"], cfg.ln_eps)
    logits = (z @ np.asarray(st["inference_model.lm_head.1.weight"], np.float32).T
              + st["inference_model.lm_head.1.bias"]) * np.asarray(repeat_penality, np.float32).reshape(1, -1)
    tok = np.argmax(logits, axis=-1).reshape(1, 1).astype(np.int32)
    return out_k, out_v, np.array([kv], np.int64), last.astype(np.float32), tok, logits.astype(np.float32)


def generate(cfg, st, conds_latent, text_ids, repeat_penality=None, max_generate_length=None, repeat_value=None,
             penalty_range=None, stop_tokens=None):
    """The per-sentence decode loop of Inference_IndexTTS_ONNX.py:716-783 (graphs B, C, D then E until a stop token).

    Returns (tokens list, last_hidden_states (n, hidden), repeat_penality after the loop).  ``repeat_penality`` is
    carried across sentences by the reference (it is initialised once at :685), hence in/out here."""
    max_generate_length = cfg.max_generate_length if max_generate_length is None else max_generate_length
    repeat_value = cfg.repeat_penalty if repeat_value is None else repeat_value
    penalty_range = cfg.penalty_range if penalty_range is None else penalty_range
    stop_tokens = [cfg.stop_mel_token] if stop_tokens is None else list(stop_tokens)
    pen = np.ones((1, cfg.mel_codes), np.float32) if repeat_penality is None else np.array(repeat_penality, np.float32)
    text_h = graph_b(cfg, st, text_ids)
    hs, gen_len = graph_c(cfg, st, [[cfg.start_mel_token]], [0])
    hs, concat_len = graph_d(np.asarray(conds_latent, np.float32), text_h, hs)
    limit = max_generate_length - int(concat_len[0])
    keys = [np.zeros((cfg.heads, cfg.head_dim, 0), np.float32)] * cfg.layers
    values = [np.zeros((cfg.heads, 0, cfg.head_dim), np.float32)] * cfg.layers
    hist, ids_len, flag = 0, int(concat_len[0]), 1
    toks, hid = [], []
    reset = 0
    n = 0
    folds = [fold_layer(cfg, st, i) for i in range(cfg.layers)]
    while n < limit:
        keys, values, kvl, last, tok, _ = graph_e(cfg, st, keys, values, hist, pen, ids_len, hs, flag, folds)
        t = int(tok[0, 0])
        toks.append(t)
        hid.append(last)
        n += 1
        if t in stop_tokens:
            break
        flag, ids_len, hist = 0, 1, int(kvl[0])
        pen[:, t] = repeat_value
        if n > penalty_range and toks[reset] != t:
            pen[:, toks[reset]] = 1.0
            reset += 1
        hs, gen_len = graph_c(cfg, st, tok, gen_len)
    return toks, np.concatenate(hid, axis=0), pen
