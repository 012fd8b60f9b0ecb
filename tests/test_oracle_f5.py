"""CPU: the numpy F5 oracle (oracle/f5_np.py) against golden vectors produced by the reference's own
module + wrapper code (tests/golden/make_golden_f5.py -> f5_small.npz)."""
import os

import numpy as np
import pytest

from mi355tts.config import F5Config
from mi355tts import weights as W
from oracle import f5_np as O


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "f5_small.npz"))


@pytest.fixture(scope="module")
def small():
    cfg = F5Config.small()
    st = W.fold_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527))
    return cfg, st


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def test_stft_b(g):
    re, im = O.stft_b(g["stft_x"])
    assert re.shape == g["stft_re"].shape == (513, 17)
    np.testing.assert_allclose(re, g["stft_re"], atol=2e-4)
    np.testing.assert_allclose(im, g["stft_im"], atol=2e-4)


def test_istft_tables_and_output(g):
    basis = O.istft_basis()
    rows = [0, 1, 7, 512, 513, 514, 700, 1025]
    np.testing.assert_allclose(basis[rows], g["istft_basis_rows"], atol=2e-7)      # closed form == fp32 pinv
    # only [n_fft/2:] is ever used (STFT_Process.py:165-166); the first samples divide by ~1e-7
    np.testing.assert_allclose(O.window_sum_inv()[512:2048], g["wsi_head"][512:], rtol=2e-6)
    np.testing.assert_allclose(O.window_sum_inv()[8:512], g["wsi_head"][8:512], rtol=1e-3)
    y = O.istft_a(g["istft_mag"], g["istft_ph"])
    assert y.shape == g["istft_y"].shape == (8 * 256,)
    np.testing.assert_allclose(y, g["istft_y"], atol=2e-5)


def test_fbank(g):
    fb = O.melscale_fbanks_htk().T
    np.testing.assert_allclose(fb[[0, 1, 50, 99]], g["fbank_rows"], atol=1e-5)   # fp64 here vs fp32 linspace in torch


def test_preprocess(g, small):
    cfg, st = small
    N = int(g["pre_N"])
    pre = O.preprocess(cfg, st, g["pre_audio"], g["pre_text_ids"], N, np.zeros((N, cfg.mel_dim), np.float32))
    assert pre["ref_signal_len"] == int(g["pre_ref_signal_len"]) == 8192 // 256 + 1
    assert np.array_equal(pre["rope_cos"], g["pre_rope_cos_q"])          # fp16-rounded tables: exact
    assert np.array_equal(pre["rope_sin"], g["pre_rope_sin_q"])
    np.testing.assert_allclose(pre["cat_mel_text"][:, :100], g["pre_cat_mel_text"][:, :100], atol=2e-3)   # log-mel
    np.testing.assert_allclose(pre["cat_mel_text"][:, 100:], g["pre_cat_mel_text"][:, 100:], atol=2e-5)   # text embed
    np.testing.assert_allclose(pre["cat_mel_text_drop"], g["pre_cat_mel_text_drop"], atol=2e-5)
    assert np.all(pre["cat_mel_text"][len(g["pre_text_ids"]):, 100:] == 0)      # filler rows are masked
    assert np.all(pre["cat_mel_text"][3, 100:] == 0)                              # pad id -1 -> filler


def test_time_tables(g, small):
    cfg, st = small
    ts, delta, texp = O.time_tables(cfg, st)
    assert len(delta) == cfg.nfe_step - 1 and abs(float(delta.sum()) - 1.0) < 1e-6
    np.testing.assert_allclose(delta, g["delta_t"], atol=1e-7)
    np.testing.assert_allclose(texp, g["time_expand"], atol=2e-5)


def test_dit_forward(g, small):
    cfg, st = small
    _, _, texp = O.time_tables(cfg, st)
    taps = {}
    pred = O.dit_forward(cfg, st, g["dit_noise"], g["pre_cat_mel_text"], g["pre_cat_mel_text_drop"], texp[2],
                         g["pre_rope_cos_q"], g["pre_rope_sin_q"], taps)
    np.testing.assert_allclose(taps["input_embed"][0], g["dit_input_embed_c"], atol=2e-5)
    np.testing.assert_allclose(taps["block.0"], g["dit_block0"], atol=5e-5)
    np.testing.assert_allclose(taps["block.1"], g["dit_block1"], atol=1e-4)
    np.testing.assert_allclose(pred, g["dit_pred_t2"], atol=1e-4)


def test_sampling_loop(g, small):
    cfg, st = small
    pre = {"noise": g["dit_noise"], "cat_mel_text": g["pre_cat_mel_text"], "cat_mel_text_drop": g["pre_cat_mel_text_drop"],
           "rope_cos": g["pre_rope_cos_q"], "rope_sin": g["pre_rope_sin_q"]}
    tables = O.time_tables(cfg, st)
    x1 = O.transformer_step(cfg, st, tables, pre["noise"], pre, 0)
    np.testing.assert_allclose(x1, g["loop_step1"], atol=5e-5)
    xf = O.sample(cfg, st, pre, tables)                      # nfe_step-1 evaluations ("NFE=32" -> 31)
    np.testing.assert_allclose(xf, g["loop_final"], atol=5e-4)


def test_decode(g, small):
    cfg, st = small
    R = int(g["pre_ref_signal_len"])
    mag, ph = O.vocos_decode(cfg, st, g["dec_in"][R:])
    np.testing.assert_allclose(mag, g["dec_mag"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(ph, g["dec_phase"], atol=1e-4)
    sig = O.decode(cfg, st, g["dec_in"], R, return_float=True)
    assert rms(sig - g["dec_float"]) < 1e-5                  # north-star bound: 1e-3 RMS
    w = O.decode(cfg, st, g["dec_in"], R)
    assert w.dtype == np.int16 and w.shape == g["dec_i16"].shape
    assert np.abs(w.astype(np.int32) - g["dec_i16"].astype(np.int32)).max() <= 2


def test_end_to_end_waveform(g, small):
    """preprocess -> sampler -> decode, against the reference chain; gate = 1e-3 RMS on [-1,1] scale."""
    cfg, st = small
    R = int(g["pre_ref_signal_len"])
    pre = {"noise": g["dit_noise"], "cat_mel_text": g["pre_cat_mel_text"], "cat_mel_text_drop": g["pre_cat_mel_text_drop"],
           "rope_cos": g["pre_rope_cos_q"], "rope_sin": g["pre_rope_sin_q"]}
    w = O.decode(cfg, st, O.sample(cfg, st, pre), R)
    err = rms((w.astype(np.float64) - g["e2e_i16"].astype(np.float64)) / 32767.0)
    assert err < 1e-3, err
    assert rms(g["e2e_i16"]) > 500


def test_host_text_helpers():
    vocab = W.synth_vocab(100)
    ids = O.list_str_to_idx(list("ab é"), vocab)
    assert ids.dtype == np.int32 and ids[2] == vocab[" "] == 0 and ids[3] == 0        # OOV -> 0
    assert O.max_duration(144000, "a" * 80, "b" * 80) == 563 + 563


def test_ref_fp16_attention_fold_is_consistent(g, small):
    """use_fp16_transformer (Export_F5.py:321-326, fp16/modules.py:467): x0.1 on q and on k, scores rounded to fp16, x100.  With
    the fold and the score scale taken from the same config the oracle stays on the reference's fp32 DiT evaluation up to the
    fp16 rounding of the scores (11 bits of a score of order one -> ~1e-3 relative in the prediction); the fold alone, without
    the x100, is a different function."""
    import dataclasses
    cfg, st = small
    cfg16 = dataclasses.replace(cfg, ref_fp16_attn=True)
    assert cfg16.to_float_array() == [cfg.cfg_strength, cfg.sway_coef, 100.0] and len(cfg.to_float_array()) == 2
    st16 = W.fold_f5(cfg16, W.synth_state(W.f5_spec(cfg16), 9527))
    k = "transformer.transformer_blocks.0.attn.to_q.weight"
    np.testing.assert_allclose(st16[k], st[k] * np.float32(0.1), rtol=3e-7)
    _, _, texp = O.time_tables(cfg, st)
    args = (g["dit_noise"], g["pre_cat_mel_text"], g["pre_cat_mel_text_drop"], texp[2], g["pre_rope_cos_q"], g["pre_rope_sin_q"])
    p16 = O.dit_forward(cfg16, st16, *args)
    ref = g["dit_pred_t2"]
    assert rms(p16 - ref) / rms(ref) < 5e-3
    assert rms(O.dit_forward(cfg, st16, *args) - ref) / rms(ref) > 2e-2        # fold without the x100: not the same function


def test_oracle_full_size_dit_evaluation_against_reference_fixture(golden_dir):
    """ONE DiT evaluation at the BASELINE shape (dim 1024, 16 heads, depth 22, N = 1126, CFG batch 2) through the oracle
    against the reference's DiT.forward run in the build container (tests/golden/make_golden_full.py -> f5_full.npz):
    pins the oracle at the real widths; the 31-step chain at this size is pinned on the GPU side against the same file."""
    import os
    from mi355tts.config import F5Config
    g = np.load(os.path.join(golden_dir, "f5_full.npz"))
    cfg = F5Config()
    st = W.fold_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527))
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, 0)
    pre = O.preprocess(cfg, st, audio[0], ids[0], N, noise[0])
    assert pre["ref_signal_len"] == int(g["ref_signal_len"]) == 563 and N == int(g["N"])
    np.testing.assert_allclose(pre["cat_mel_text"][:, :100], g["pre_cat_mel_text"][:, :100], atol=3e-3)
    np.testing.assert_allclose(pre["cat_mel_text"][:, 100:], g["pre_cat_mel_text"][:, 100:], atol=1e-4)
    tables = O.time_tables(cfg, st)
    pred = O.dit_forward(cfg, st, pre["noise"], pre["cat_mel_text"], pre["cat_mel_text_drop"], tables[2][7], pre["rope_cos"],
                         pre["rope_sin"])
    ref = g["dit_pred_t7"]
    assert pred.shape == ref.shape == (2, N, cfg.mel_dim)
    rel = float(np.sqrt(np.mean((pred - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    assert rel < 1e-4, rel


def test_bigvgan_type_mel_oracle_against_reference_fixture(golden_dir):
    """oracle/f5_np.py bigvgan_mel (modules.py:30-72) against the output of the reference's own function on 1 s of the bench
    prompt and 1 s of zh.wav (tests/golden/make_golden_bigvgan_mel.py); the slaney basis against its closed-form landmarks."""
    g = np.load(os.path.join(golden_dir, "f5_bigvgan_mel.npz"))
    assert float(g["basis_check_max_abs"]) < 1e-7          # generator: restated librosa basis == transformers' slaney/slaney bank
    for name, tol in (("syn", 2e-5), ("zh", 5e-4)):
        ref = g[name + "_logmel"].T
        got = O.bigvgan_mel(g[name + "_pcm"].astype(np.float32) * np.float32(1.0 / 32768.0))
        m = ref > np.log(2e-5)
        assert got.shape == ref.shape and np.abs(got - ref)[m].max() < tol
    b = O.mel_basis_slaney()
    assert b.shape == (100, 513) and (b >= 0).all() and abs(float(b[50].sum() * (12000.0 / 512)) - 1.0) < 0.02     # slaney norm: unit-area triangles (a mid band spans many bins)
