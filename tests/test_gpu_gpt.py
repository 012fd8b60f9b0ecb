"""GPU parity: the HIP IndexTTS GPT path (graphs B, C, E and the on-device decode loop, through the C-ABI) against the
golden vectors produced by the reference wrapper classes over Hugging Face GPT2Block modules, and against the numpy
oracle on larger seeded cases.

Tolerances
  fp32: exact-fp32 arithmetic (fp32 MFMA for the prompt pass, fp32 FMAs in the GEMV / attention kernels); only the
        summation order differs: <= 2e-4 abs on O(1) hidden states, greedy tokens identical.
  fp16 / bf16: weights, KV cache and GEMV inputs are rounded to 16 bits (like the reference's whole-graph fp16 cast,
        IndexTTS/Export_IndexTTS.py fp16 branch); teacher-forced hidden states within 4e-2 / 2e-1 abs.
"""
import os

import numpy as np
import pytest

from mi355tts import weights as W
from mi355tts import _lib
from mi355tts.config import IndexGPTConfig
from mi355tts.indextts import IndexGPT
from oracle import gpt_np as O

pytestmark = pytest.mark.gpu
SEED = 9527


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "indextts_gpt.npz"))


@pytest.fixture(scope="module")
def small():
    cfg = IndexGPTConfig.small()
    st = W.synth_state(W.gpt_spec(cfg), SEED)
    return cfg, st


@pytest.fixture(scope="module")
def eng(small):
    cfg, st = small
    e = IndexGPT(cfg, st, dtype="f32")
    yield e
    e.close()


def test_graph_b_c_d_golden(eng, g):
    tb = eng.text_embed(g["text_ids"])
    np.testing.assert_allclose(tb, g["B_text_hidden"], rtol=0, atol=1e-6)
    hc, gl = eng.mel_embed([[eng.cfg.start_mel_token]], [0])
    np.testing.assert_allclose(hc, g["C_hidden_0"], rtol=0, atol=1e-6)
    assert int(gl[0]) == 1
    d, n = eng.concat(g["conds_latent"], tb, hc)
    np.testing.assert_allclose(d, g["D_hidden"], rtol=0, atol=1e-6)
    assert int(n[0]) == int(g["D_len"][0])


def test_graph_e_prompt_pass_golden(eng, g):
    eng.reset()
    kv, last, tok = eng.step(g["D_hidden"], np.ones((1, eng.cfg.mel_codes), np.float32), attention_mask=1)
    assert int(kv[0]) == 13 and eng.history_len == 13
    np.testing.assert_allclose(last, g["E0_last_hidden"], rtol=0, atol=2e-4)
    assert int(tok[0, 0]) == int(g["gen_tokens"][0])
    k0, _ = eng.kv_read(0)
    _, v1 = eng.kv_read(1)
    np.testing.assert_allclose(k0, g["E0_key0"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(v1, g["E0_value1"], rtol=0, atol=2e-4)


def test_graph_e_single_step_with_given_cache_golden(eng, g):
    eng.kv_write(list(g["S_keys_in"]), list(g["S_values_in"]))
    hist = g["S_keys_in"].shape[3]
    assert eng.history_len == hist
    kv, last, tok = eng.step(g["S_hidden_in"], g["S_pen"], attention_mask=0)
    assert int(kv[0]) == hist + 1
    np.testing.assert_allclose(last, g["S_last_hidden"], rtol=0, atol=2e-4)
    assert int(tok[0, 0]) == int(g["S_token"][0, 0])


def test_driver_loop_step_by_step_golden(eng, g):
    """Inference_IndexTTS_ONNX.py:752-783 driven from the host, one mi_gpt_step per token (the drop-in shape)."""
    cfg = eng.cfg
    rep, prange = float(g["gen_params"][0]), int(g["gen_params"][1])
    eng.reset()
    pen = np.ones((1, cfg.mel_codes), np.float32)
    hs, gen_len = g["D_hidden"], np.array([1])
    flag, toks, reset = 1, [], 0
    for n in range(len(g["gen_tokens"])):
        kv, last, tok = eng.step(hs, pen, attention_mask=flag)
        t = int(tok[0, 0])
        toks.append(t)
        np.testing.assert_allclose(last[0], g["gen_hidden"][n], rtol=0, atol=3e-4)
        flag = 0
        pen[:, t] = rep
        if n + 1 > prange and toks[reset] != t:
            pen[:, toks[reset]] = 1.0
            reset += 1
        hs, gen_len = eng.mel_embed(tok, gen_len)
    assert toks == [int(x) for x in g["gen_tokens"]]
    np.testing.assert_array_equal(pen, g["gen_penalty"])
    k0, _ = eng.kv_read(0)
    np.testing.assert_allclose(k0, g["gen_key0"], rtol=0, atol=3e-4)


def test_generate_on_device_golden(eng, g):
    rep, prange = float(g["gen_params"][0]), int(g["gen_params"][1])
    n = len(g["gen_tokens"])
    toks, hid, pen = eng.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + n, stop_tokens=[],
                                  repeat_value=rep, penalty_range=prange,
                                  repeat_penality=np.ones((1, eng.cfg.mel_codes), np.float32))
    assert toks.tolist() == [int(x) for x in g["gen_tokens"]]
    np.testing.assert_allclose(hid, g["gen_hidden"], rtol=0, atol=3e-4)
    np.testing.assert_array_equal(pen, g["gen_penalty"])
    # second run replays the captured decode step: same answer
    toks2, hid2, _ = eng.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + n, stop_tokens=[],
                                  repeat_value=rep, penalty_range=prange,
                                  repeat_penality=np.ones((1, eng.cfg.mel_codes), np.float32))
    assert toks2.tolist() == toks.tolist()
    np.testing.assert_array_equal(hid2, hid)


def test_generate_stops_at_stop_token(eng, small, g):
    rep, prange = float(g["gen_params"][0]), int(g["gen_params"][1])
    stop = int(g["gen_tokens"][4])
    first = [int(x) for x in g["gen_tokens"]].index(stop)
    toks, hid, pen = eng.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + 12, stop_tokens=[stop, 1000],
                                  repeat_value=rep, penalty_range=prange,
                                  repeat_penality=np.ones((1, eng.cfg.mel_codes), np.float32))
    assert toks.tolist() == [int(x) for x in g["gen_tokens"][: first + 1]]
    assert hid.shape == (first + 1, eng.cfg.hidden)
    # the stop token itself is not penalised (the reference breaks before the update, :761-762): same as the oracle
    cfg, st = small
    o_toks, _, o_pen = O.generate(cfg, st, g["conds_latent"], g["text_ids"], max_generate_length=13 + 12,
                                  repeat_value=rep, penalty_range=prange, stop_tokens=[stop, 1000])
    assert toks.tolist() == o_toks
    np.testing.assert_array_equal(pen, o_pen)


def test_generate_limit_and_empty(eng, g):
    toks, hid, _ = eng.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + 3, stop_tokens=[])
    assert len(toks) == 3 and hid.shape[0] == 3
    toks, hid, _ = eng.generate(g["conds_latent"], g["text_ids"], max_generate_length=13, stop_tokens=[])
    assert len(toks) == 0 and hid.shape[0] == 0


def test_penalty_carries_across_sentences(small, g):
    cfg, st = small
    e = IndexGPT(cfg, st, dtype="f32")
    t1, _, p1 = e.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + 6, stop_tokens=[])
    assert np.array_equal(e.repeat_penality, p1) and (p1 != 1).any()
    o1, _, op1 = O.generate(cfg, st, g["conds_latent"], g["text_ids"], max_generate_length=13 + 6, stop_tokens=[])
    assert t1.tolist() == o1
    t2, _, p2 = e.generate(g["conds_latent"], g["text_ids"][:, :4], max_generate_length=11 + 6, stop_tokens=[])
    o2, _, op2 = O.generate(cfg, st, g["conds_latent"], g["text_ids"][:, :4], repeat_penality=op1,
                            max_generate_length=11 + 6, stop_tokens=[])
    assert t2.tolist() == o2
    np.testing.assert_array_equal(p2, op2)
    e.close()


@pytest.mark.parametrize("hidden,inner", [(256, 1024), (320, 1352), (704, 2816)])
@pytest.mark.parametrize("dtype,tol", [("f32", 3e-4), ("f16", 4e-2), ("bf16", 2.5e-1)])
def test_medium_model_vs_oracle(dtype, tol, hidden, inner):
    """3 layers, 75-row prompt (MFMA GEMM path with a ragged tile), then teacher-forced single steps (the decode-step GEMV
    kernel) against the oracle.  The widths walk the GEMV's compile-time K iterations and its padded lanes: hidden 256 (one
    iteration at 16 bits, none padded), 320 (lanes 40-63 padded; inner 1352 = 2.64 iterations), 704 / 2816 (1.4 / 5.5
    iterations at 16 bits, 2.75 / 11 in fp32); 301 lm_head rows end inside a block."""
    cfg = IndexGPTConfig(hidden=hidden, layers=3, heads=hidden // 64, inner=inner, mel_codes=301, text_tokens=64, max_mel_pos=80,
                         max_text_pos=80, max_seq=160, start_mel_token=299, stop_mel_token=300, max_generate_length=120)
    st = W.synth_state(W.gpt_spec(cfg), 77)
    e = IndexGPT(cfg, st, dtype=dtype)
    conds = W.synth_normal(5, "conds", (1, 32, cfg.hidden), std=0.5)
    text = (np.arange(40, dtype=np.int32) * 7 % 60 + 2)[None]
    tb = e.text_embed(text)
    mh, gl = e.mel_embed(cfg.start_mel_token, 0)
    prompt, n = e.concat(conds, tb, mh)
    assert int(n[0]) == 75
    np.testing.assert_allclose(prompt, O.graph_d(conds, O.graph_b(cfg, st, text), O.graph_c(cfg, st, [[299]], [0])[0])[0],
                               rtol=0, atol=1e-6)
    pen = W.synth_normal(6, "pen", (1, cfg.mel_codes), std=0.1, mean=1.0)
    keys = [np.zeros((cfg.heads, 64, 0), np.float32)] * cfg.layers
    vals = [np.zeros((cfg.heads, 0, 64), np.float32)] * cfg.layers
    e.reset()
    kv, last, tok, logits = e.step(prompt, pen, attention_mask=1, return_logits=True)
    keys, vals, okv, olast, otok, ologits = O.graph_e(cfg, st, keys, vals, 0, pen, 75, prompt, 1)
    assert int(kv[0]) == int(okv[0]) == 75
    np.testing.assert_allclose(last, olast, rtol=0, atol=tol)
    np.testing.assert_allclose(logits * pen, ologits, rtol=0, atol=tol * 4)
    if dtype == "f32":
        assert int(tok[0, 0]) == int(otok[0, 0])
    hist = 75
    for s in range(6):                                    # teacher-forced with the oracle's tokens
        hs, gl = e.mel_embed(otok, gl)
        kv, last, tok, logits = e.step(hs, pen, attention_mask=0, return_logits=True)
        keys, vals, okv, olast, otok2, ologits = O.graph_e(cfg, st, keys, vals, hist, pen, 1, hs, 0)
        hist += 1
        np.testing.assert_allclose(last, olast, rtol=0, atol=tol)
        np.testing.assert_allclose(logits * pen, ologits, rtol=0, atol=tol * 4)
        if dtype == "f32":
            assert int(tok[0, 0]) == int(otok2[0, 0])
        # argmax consistency with the engine's own logits (first index on ties)
        assert int(tok[0, 0]) == int(np.argmax(logits * pen))
        otok = otok2
    k1, v1 = e.kv_read(1)
    np.testing.assert_allclose(k1, keys[1], rtol=0, atol=tol)
    np.testing.assert_allclose(v1, vals[1], rtol=0, atol=tol)
    e.close()


def test_long_history_attention(small):
    """kv length > 512 (more than one pass of the 512-thread score loop) on the small model."""
    cfg0, _ = small
    cfg = IndexGPTConfig(**{**cfg0.__dict__, "max_seq": 640, "max_mel_pos": 640})
    st = W.synth_state(W.gpt_spec(cfg), 3)
    e = IndexGPT(cfg, st, dtype="f32")
    hist = 600
    keys = [W.synth_normal(9, f"k{i}", (cfg.heads, 64, hist), std=0.6) for i in range(cfg.layers)]
    vals = [W.synth_normal(9, f"v{i}", (cfg.heads, hist, 64), std=0.6) for i in range(cfg.layers)]
    e.kv_write(keys, vals)
    hs = W.synth_normal(9, "hs", (1, 1, cfg.hidden), std=0.7)
    kv, last, tok = e.step(hs, None, attention_mask=0)
    _, _, _, olast, otok, _ = O.graph_e(cfg, st, keys, vals, hist, np.ones((1, cfg.mel_codes), np.float32), 1, hs, 0)
    np.testing.assert_allclose(last, olast, rtol=0, atol=3e-4)
    assert int(tok[0, 0]) == int(otok[0, 0]) and int(kv[0]) == hist + 1
    e.close()


def test_graph_replay_equals_eager(small, g, monkeypatch):
    cfg, st = small
    a = IndexGPT(cfg, st, dtype="f16")
    monkeypatch.setenv("MI355TTS_NO_GRAPH", "1")
    b = IndexGPT(cfg, st, dtype="f16")
    monkeypatch.delenv("MI355TTS_NO_GRAPH")
    for _ in range(2):
        ta, ha, _ = a.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + 20, stop_tokens=[],
                               repeat_penality=np.ones((1, cfg.mel_codes), np.float32))
        tb, hb, _ = b.generate(g["conds_latent"], g["text_ids"], max_generate_length=13 + 20, stop_tokens=[],
                               repeat_penality=np.ones((1, cfg.mel_codes), np.float32))
        assert ta.tolist() == tb.tolist() and len(ta) == 20
        np.testing.assert_array_equal(ha, hb)
    a.close()
    b.close()


def test_errors(eng, g):
    with pytest.raises(_lib.MiError):
        eng.text_embed(np.array([[eng.cfg.text_tokens]], np.int32))          # id out of range
    with pytest.raises(_lib.MiError):
        eng.text_embed(np.zeros((1, eng.cfg.max_text_pos), np.int32))         # longer than the position table
    with pytest.raises(_lib.MiError):
        eng.mel_embed(eng.cfg.mel_codes, 0)
    eng.reset()
    with pytest.raises(_lib.MiError):
        eng.step(np.zeros((1, eng.cfg.max_seq + 1, eng.cfg.hidden), np.float32), None, 1)
    with pytest.raises(_lib.MiError):
        eng.generate_from_prompt(np.zeros((1, 10, eng.cfg.hidden), np.float32), eng.cfg.max_seq)
    with pytest.raises(ValueError):
        eng.step(np.zeros((1, 2, 7), np.float32))
    with pytest.raises(ValueError):
        eng.kv_write([np.zeros((2, 64, 3))], [np.zeros((2, 3, 64))])


# ---------------------------------------------------------------------------------------------------------------
# batched decode (engine extension): every sentence must come out exactly as if it had been decoded alone
# ---------------------------------------------------------------------------------------------------------------
def _prompt(e, cfg, seed, n_text, n_cond=4):
    conds = W.synth_normal(seed, "conds", (1, n_cond, cfg.hidden), std=0.5)
    text = (np.arange(n_text, dtype=np.int32) * 5 + seed) % (cfg.text_tokens - 2) + 2
    mh, _ = e.mel_embed(cfg.start_mel_token, 0)
    p, _ = e.concat(conds, e.text_embed(text), mh)
    return conds, text, p


def test_generate_batch_equals_reference_loop_per_sentence(small):
    cfg0, st = small
    cfg = IndexGPTConfig(**{**cfg0.__dict__, "max_batch": 4})
    e = IndexGPT(cfg, st, dtype="f32")
    items = [_prompt(e, cfg, 1, 6), _prompt(e, cfg, 2, 3), _prompt(e, cfg, 3, 9)]
    limits = [14, 9, 11]
    # sentence 1 stops early on a token the oracle is known to emit
    o_free, _, _ = O.generate(cfg, st, items[1][0], items[1][1], max_generate_length=items[1][2].shape[1] + 9, stop_tokens=[])
    stop = o_free[4]
    res, pen = e.generate_batch([it[2] for it in items], limits, stop_tokens=[stop])
    for b, (conds, text, p) in enumerate(items):
        ot, oh, op = O.generate(cfg, st, conds, text, max_generate_length=p.shape[1] + limits[b], stop_tokens=[stop])
        assert res[b][0].tolist() == ot, b
        np.testing.assert_allclose(res[b][1], oh, rtol=0, atol=3e-4)
        np.testing.assert_array_equal(pen[b:b + 1], op)
    assert len(res[1][0]) <= 5 or stop in res[1][0].tolist()
    # and again through the captured graph, with one idle slot (max_new = 0)
    res2, _ = e.generate_batch([it[2] for it in items], [limits[0], 0, limits[2]], stop_tokens=[stop])
    assert res2[0][0].tolist() == res[0][0].tolist() and len(res2[1][0]) == 0 and res2[2][0].tolist() == res[2][0].tolist()
    # the single-sentence API still works on the same handle afterwards
    t, _, _ = e.generate_from_prompt(items[0][2], limits[0], stop_tokens=[stop], repeat_penality=np.ones((1, cfg.mel_codes), np.float32))
    assert t.tolist() == res[0][0].tolist()
    with pytest.raises(ValueError):
        e.generate_batch([it[2] for it in items] * 2, limits * 2)
    e.close()


@pytest.mark.parametrize("nb", [2, 5, 9, 16])
def test_generate_batch_template_widths(nb):
    cfg = IndexGPTConfig(hidden=256, layers=2, heads=4, inner=1024, mel_codes=301, text_tokens=64, max_mel_pos=80,
                         max_text_pos=80, max_seq=96, max_batch=16, start_mel_token=299, stop_mel_token=300)
    st = W.synth_state(W.gpt_spec(cfg), 11)
    e = IndexGPT(cfg, st, dtype="f32")
    items = [_prompt(e, cfg, 10 + b, 3 + (b * 7) % 11, n_cond=8) for b in range(nb)]
    limits = [6 + (b * 5) % 9 for b in range(nb)]
    res, pen = e.generate_batch([it[2] for it in items], limits, stop_tokens=[])
    for b in range(nb):
        t, h, p = e.generate_from_prompt(items[b][2], limits[b], stop_tokens=[],
                                         repeat_penality=np.ones((1, cfg.mel_codes), np.float32))
        assert res[b][0].tolist() == t.tolist(), b
        np.testing.assert_allclose(res[b][1], h, rtol=0, atol=2e-4)
        np.testing.assert_array_equal(pen[b:b + 1], p)
    e.close()


def test_generate_batch_f16_tracks_single(small):
    cfg0, st = small
    cfg = IndexGPTConfig(**{**cfg0.__dict__, "max_batch": 4})
    e = IndexGPT(cfg, st, dtype="f16")
    items = [_prompt(e, cfg, 1, 6), _prompt(e, cfg, 2, 3), _prompt(e, cfg, 3, 9), _prompt(e, cfg, 4, 5)]
    res, _ = e.generate_batch([it[2] for it in items], [10] * 4, stop_tokens=[])
    agree = 0
    for b in range(4):
        t, h, _ = e.generate_from_prompt(items[b][2], 10, stop_tokens=[], repeat_penality=np.ones((1, cfg.mel_codes), np.float32))
        assert res[b][0][0] == t[0]                                 # the prompt pass is the same code path
        np.testing.assert_array_equal(res[b][1][0], h[0])
        k = 0
        while k < 10 and res[b][0][k] == t[k]:
            np.testing.assert_allclose(res[b][1][k], h[k], rtol=0, atol=5e-2)
            k += 1
        agree += k
    assert agree >= 30          # 16-bit rounding differs between the fused-LN GEMV and the batched GEMV; ties are rare
    e.close()


def _teacher_forced_hidden(cfg, st, prompt, toks):
    """Oracle hidden states when it is fed the ENGINE's tokens (no divergence: 16-bit engines may legitimately pick a
    different near-tie token than the fp32 oracle)."""
    keys = [np.zeros((cfg.heads, 64, 0), np.float32)] * cfg.layers
    vals = [np.zeros((cfg.heads, 0, 64), np.float32)] * cfg.layers
    pen = np.ones((1, cfg.mel_codes), np.float32)
    folds = [O.fold_layer(cfg, st, i) for i in range(cfg.layers)]
    keys, vals, kvl, last, _, logits = O.graph_e(cfg, st, keys, vals, 0, pen, prompt.shape[1], prompt, 1, folds)
    out, lg = [last], [logits]
    gl = np.array([1])
    for t in toks[:-1]:
        hs, gl = O.graph_c(cfg, st, [[int(t)]], gl)
        keys, vals, kvl, last, _, logits = O.graph_e(cfg, st, keys, vals, int(kvl[0]), pen, 1, hs, 0, folds)
        out.append(last); lg.append(logits)
    return np.concatenate(out, 0), np.concatenate(lg, 0)


@pytest.mark.parametrize("dtype,tol", [("f16", 4e-2), ("bf16", 2.5e-1)])
@pytest.mark.parametrize("nb", [3, 9, 16])
def test_generate_batch_matrix_core_path(dtype, tol, nb):
    """16-bit engines with >= 9 sentences (threshold lowered to 3 here) run the decode-step linears as v_mfma_f32_16x16x32 skinny GEMMs (hidden 256 and
    320: one and five 64-wide K blocks per wave; inner 1280 takes the 8-way K split).  Checked against the oracle fed with
    the engine's own tokens."""
    _lib.set_option("gpt_mfma_min", 3)
    for hidden, heads, inner in ((256, 4, 1024), (320, 5, 2560)):
        cfg = IndexGPTConfig(hidden=hidden, layers=2, heads=heads, inner=inner, mel_codes=301, text_tokens=64, max_mel_pos=80,
                             max_text_pos=80, max_seq=96, max_batch=16, start_mel_token=299, stop_mel_token=300)
        st = W.synth_state(W.gpt_spec(cfg), 13)
        e = IndexGPT(cfg, st, dtype=dtype)
        items = [_prompt(e, cfg, 20 + b, 3 + (b * 5) % 9, n_cond=6) for b in range(nb)]
        limits = [5 + (b * 3) % 6 for b in range(nb)]
        res, _ = e.generate_batch([it[2] for it in items], limits, stop_tokens=[], repeat_value=1.0)
        for b in (0, nb // 2, nb - 1):
            toks, hid = res[b]
            assert len(toks) == limits[b]
            ohid, ologits = _teacher_forced_hidden(cfg, st, items[b][2], toks)
            np.testing.assert_allclose(hid, ohid, rtol=0, atol=tol)
            # the chosen token is (near-)maximal in the oracle's logits
            for k, t in enumerate(toks):
                assert ologits[k, t] >= ologits[k].max() - 6 * tol, (b, k)
        e.close()
    _lib.set_option("gpt_mfma_min", 9)


# ---------------------------------------------------------------------------------------------
# BASELINE.json size (IndexTTS-1.5 GPT: 24 layers x 1280, 20 heads, inner 5120, f16 engine): the reference loop's
# oracle is too slow for a 256-token decode, so the full size is held by
#   * a TEACHER-FORCED oracle pass over the first few engine tokens (hidden states within the stated f16 bar),
#   * batch == single: the same prompt in every slot of mi_gpt_generate_batch decodes the single-sentence tokens
#     (batched GEMV / matrix-core skinny GEMM against the fused-LN GEMV, at the real shapes), and
#   * replay determinism (the per-token hipGraph).
# ---------------------------------------------------------------------------------------------
def test_full_size_gpt_teacher_forced_and_batch_equals_single():
    cfg = IndexGPTConfig()
    assert (cfg.layers, cfg.hidden, cfg.heads) == (24, 1280, 20)
    cfg.max_batch = 4
    st = W.synth_state(W.gpt_spec(cfg), SEED, fast=True)
    e = IndexGPT(cfg, st, dtype="f16")
    conds, text, p = _prompt(e, cfg, 3, 12, n_cond=32)
    ones = np.ones((1, cfg.mel_codes), np.float32)
    n_new = 16
    t1, h1, _ = e.generate_from_prompt(p, n_new, stop_tokens=[], repeat_penality=ones.copy())
    t2, h2, _ = e.generate_from_prompt(p, n_new, stop_tokens=[], repeat_penality=ones.copy())     # replayed graph
    assert len(t1) == n_new and t1.tolist() == t2.tolist()
    np.testing.assert_array_equal(h1, h2)
    # oracle fed the engine's first tokens
    n_chk = 4
    oh, _ = _teacher_forced_hidden(cfg, st, p, t1[:n_chk])
    assert np.isfinite(h1).all() and float(np.abs(oh).max()) > 0.1
    assert float(np.abs(h1[:n_chk] - oh).max()) < 4e-2 * max(1.0, float(np.abs(oh).max()))
    # the same sentence in all four slots
    res, _ = e.generate_batch([p] * 4, [n_new] * 4, stop_tokens=[])
    for b in range(4):
        assert res[b][0].tolist() == res[0][0].tolist()             # slots are independent and identical
        k = 0
        while k < n_new and res[b][0][k] == t1[k]:
            k += 1
        assert k >= n_new // 2, (b, k)                                # 16-bit rounding order differs between the two GEMV forms
    e.close()


def test_repeat_penalty_change_on_one_handle_reaches_the_replayed_graph():
    """The decode step is captured into a hipGraph on first use; REPEAT_PENALITY is a device scalar so that a later call with
    another value is honoured by every replayed token (it used to be baked into the captured kernel arguments)."""
    from mi355tts.config import IndexGPTConfig
    from mi355tts.indextts import IndexGPT
    from oracle import gpt_np as G
    cfg = IndexGPTConfig.small()
    st = W.synth_state(W.gpt_spec(cfg), 9527)
    conds = W.synth_normal(9527, "rp.conds", (1, 4, cfg.hidden), std=0.5)
    text = np.array([[5, 17, 3, 22, 9, 30]], np.int32)
    gpt = IndexGPT(cfg, st, dtype="f32")
    outs = {}
    for rep in (0.7, 0.2, 0.7):
        ones = np.ones((1, cfg.mel_codes), np.float32)
        toks, hid, _ = gpt.generate(conds, text, max_generate_length=13 + 14, stop_tokens=[], repeat_value=rep, repeat_penality=ones)
        otoks, _, _ = G.generate(cfg, st, conds, text, max_generate_length=13 + 14, stop_tokens=[], repeat_value=rep)
        assert toks.tolist() == otoks, rep
        outs.setdefault(rep, toks.tolist())
        assert outs[rep] == toks.tolist()
    assert outs[0.7] != outs[0.2]                 # the two penalties really give different token streams
    gpt.close()
