"""oracle/gpt_np.py against the golden vectors produced by the reference IndexTTS_B/C/D/E wrappers over Hugging Face
GPT2Block modules (tests/golden/make_golden_gpt.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, ROOT)

from mi355tts import weights as W                 # noqa: E402
from mi355tts.config import IndexGPTConfig        # noqa: E402
from oracle import gpt_np as G                    # noqa: E402

SEED = 9527


@pytest.fixture(scope="module")
def env():
    cfg = IndexGPTConfig.small()
    st = W.synth_state(W.gpt_spec(cfg), SEED)
    g = np.load(os.path.join(ROOT, "tests", "golden", "indextts_gpt.npz"))
    return cfg, st, g


def test_graphs_b_c_d(env):
    cfg, st, g = env
    tb = G.graph_b(cfg, st, g["text_ids"])
    np.testing.assert_allclose(tb, g["B_text_hidden"], rtol=0, atol=1e-6)
    hc, gl = G.graph_c(cfg, st, [[cfg.start_mel_token]], [0])
    np.testing.assert_allclose(hc, g["C_hidden_0"], rtol=0, atol=1e-6)
    assert int(gl[0]) == 1
    d, n = G.graph_d(g["conds_latent"], tb, hc)
    np.testing.assert_allclose(d, g["D_hidden"], rtol=0, atol=1e-6)
    assert int(n[0]) == int(g["D_len"][0]) == 13


def test_graph_e_prefill(env):
    cfg, st, g = env
    keys = [np.zeros((cfg.heads, cfg.head_dim, 0), np.float32)] * cfg.layers
    vals = [np.zeros((cfg.heads, 0, cfg.head_dim), np.float32)] * cfg.layers
    pen = np.ones((1, cfg.mel_codes), np.float32)
    k, v, kvl, last, tok, _ = G.graph_e(cfg, st, keys, vals, 0, pen, 13, g["D_hidden"], 1)
    assert int(kvl[0]) == 13
    np.testing.assert_allclose(last, g["E0_last_hidden"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(k[0], g["E0_key0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(v[1], g["E0_value1"], rtol=0, atol=2e-5)
    assert int(tok[0, 0]) == int(g["gen_tokens"][0])


def test_graph_e_single_step_with_cache_and_penalty(env):
    cfg, st, g = env
    keys, vals = list(g["S_keys_in"]), list(g["S_values_in"])
    hist = keys[0].shape[2]
    _, _, kvl, last, tok, _ = G.graph_e(cfg, st, keys, vals, hist, g["S_pen"], 1, g["S_hidden_in"], 0)
    assert int(kvl[0]) == hist + 1
    np.testing.assert_allclose(last, g["S_last_hidden"], rtol=0, atol=2e-5)
    assert int(tok[0, 0]) == int(g["S_token"][0, 0])


def test_generate_loop_matches_reference_driver(env):
    cfg, st, g = env
    rep, prange = float(g["gen_params"][0]), int(g["gen_params"][1])
    n = len(g["gen_tokens"])
    toks, hid, pen = G.generate(cfg, st, g["conds_latent"], g["text_ids"], max_generate_length=13 + n,
                                repeat_value=rep, penalty_range=prange, stop_tokens=[])
    assert toks == [int(x) for x in g["gen_tokens"]]
    np.testing.assert_allclose(hid, g["gen_hidden"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pen, g["gen_penalty"], rtol=0, atol=0)


def test_fold_matches_per_head_surgery(env):
    """weights.fold_gpt (what the engine blob holds) == the wrapper's per-head tensors."""
    cfg, st, _ = env
    fs = W.fold_gpt(cfg, st)
    h, H, D = cfg.hidden, cfg.heads, cfg.head_dim
    for i in range(cfg.layers):
        f = G.fold_layer(cfg, st, i)
        w = fs[f"inference_model.transformer.h.{i}.attn.c_attn.weight"]          # (3h, h) rows = outputs
        b = fs[f"inference_model.transformer.h.{i}.attn.c_attn.bias"]
        for j, nm in enumerate("qkv"):
            np.testing.assert_array_equal(w[j * h:(j + 1) * h].reshape(H, D, h).transpose(0, 2, 1), f["w" + nm])
            np.testing.assert_array_equal(b[j * h:(j + 1) * h].reshape(H, 1, D), f["b" + nm])
        wo = fs[f"inference_model.transformer.h.{i}.attn.c_proj.weight"]          # (out, in = head*D + d)
        np.testing.assert_array_equal(wo.reshape(h, H, D).transpose(1, 2, 0), f["wo"])
    assert W.pack_gpt(cfg, st).size == sum(int(np.prod(s)) for _, s, _ in W.gpt_spec(cfg))
