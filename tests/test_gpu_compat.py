"""GPU: the reference driver's call sequence through the onnxruntime-shaped façade
(`import mi355tts.ort_compat as onnxruntime`), F5-TTS-ONNX-Inference.py:173-311 and
Export_BigVGAN.py:153-175."""
import os

import numpy as np
import pytest

from mi355tts.config import BigVGANConfig, F5Config
from mi355tts import weights as W
from mi355tts import ort_compat as onnxruntime
from mi355tts import text as T

pytestmark = pytest.mark.gpu


def test_f5_driver_sequence_through_facade(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    cfg = F5Config.small()
    wfile = tmp_path / "f5_weights.npy"
    np.save(wfile, W.pack_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527)))
    paths = {k: onnxruntime.save_model(str(tmp_path / f"{k}.mi355.json"), k, cfg, str(wfile), "f32")
             for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")}
    onnxruntime.set_seed(9527)
    so = onnxruntime.SessionOptions()
    so.add_session_config_entry("session.set_denormal_as_zero", "1")
    A = onnxruntime.InferenceSession(paths["F5_Preprocess"], sess_options=so, providers=["CPUExecutionProvider"])
    B = onnxruntime.InferenceSession(paths["F5_Transformer"], sess_options=so, providers=[], provider_options=None)
    Cc = onnxruntime.InferenceSession(paths["F5_Decode"], sess_options=so)
    in_A, out_A = [a.name for a in A.get_inputs()], [a.name for a in A.get_outputs()]
    in_B, out_B = [a.name for a in B.get_inputs()], [a.name for a in B.get_outputs()]
    audio = g["pre_audio"].reshape(1, 1, -1)
    text_ids = g["pre_text_ids"].reshape(1, -1)
    max_duration = np.array([int(g["pre_N"])], dtype=np.int64)
    noise, cq, sq, ck, sk, cmt, cmtd, rsl = A.run(out_A, {in_A[0]: audio, in_A[1]: text_ids, in_A[2]: max_duration})
    assert noise.shape == (1, int(g["pre_N"]), 100) and abs(float(noise.std()) - 1.0) < 0.1   # seeded N(0,1)
    noise = g["dit_noise"][None].copy()                          # inject the golden noise from here on
    time_step = np.array([0], dtype=np.int32)
    for i in range(0, cfg.nfe_step - 1, 1):
        noise, time_step = B.run([out_B[0], out_B[1]], {in_B[0]: noise, in_B[1]: cq, in_B[2]: sq, in_B[3]: ck, in_B[4]: sk,
                                                      in_B[5]: cmt, in_B[6]: cmtd, in_B[7]: time_step})
    wav = Cc.run([Cc.get_outputs()[0].name], {Cc.get_inputs()[0].name: noise, Cc.get_inputs()[1].name: rsl})[0]
    assert wav.dtype == np.int16 and wav.shape == (1, 1, g["e2e_i16"].shape[0])
    err = np.sqrt(np.mean(((wav[0, 0].astype(np.float64) - g["e2e_i16"]) / 32767.0) ** 2))
    assert err < 5e-4, err
    with pytest.raises(onnxruntime.InvalidArgument):
        B.run(None, {in_B[0]: noise})
    with pytest.raises(onnxruntime.InvalidArgument):
        A.run(out_A, {in_A[0]: audio.astype(np.float32), in_A[1]: text_ids, in_A[2]: max_duration})
    # one-call convenience == A -> loop -> C
    eng = A._eng
    w2 = eng.synthesize(g["pre_audio"][None], g["pre_text_ids"][None], int(g["pre_N"]), noise=g["dit_noise"][None])
    assert np.array_equal(w2, wav)


def test_bigvgan_session_like_the_reference_smoke_run(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "bigvgan_small.npz"))
    cfg = BigVGANConfig.small()
    wfile = tmp_path / "bv.npy"
    np.save(wfile, W.pack_bigvgan(cfg, W.synth_state(W.bigvgan_spec(cfg), 9527)))
    path = onnxruntime.save_model(str(tmp_path / "BigVGAN.mi355.json"), "BigVGAN", cfg, str(wfile), "f32")
    sess = onnxruntime.InferenceSession(path, sess_options=onnxruntime.SessionOptions(), providers=[], provider_options=None)
    assert "MI355X" in sess.get_providers()[0]
    dt = np.float16 if "float16" in sess._inputs_meta[0].type else np.float32
    dummy = onnxruntime.OrtValue.ortvalue_from_numpy(np.ones((1, sess._inputs_meta[0].shape[1], 12), dtype=dt), "cpu", 0)
    out = sess.run_with_ort_values([sess.get_outputs()[0].name], {sess.get_inputs()[0].name: dummy})
    w = out[0].numpy()
    assert np.abs(w.astype(np.int32) - g["gen_i16_ones"].astype(np.int32)).max() <= 1


def test_synthesize_convenience_text_in_wave_out(tmp_path):
    from mi355tts.f5 import F5Engine
    cfg = F5Config.small()
    eng = F5Engine(cfg, W.synth_state(W.f5_spec(cfg), 9527))
    vocab = W.synth_vocab(cfg.text_num_embeds)
    audio = (3000 * np.sin(np.arange(6000) * 0.05)).astype(np.int16)
    w = onnxruntime.synthesize(eng, audio, "ABC AB", "AB CA", vocab, seed=1)
    N = T.max_duration(6000, "ABC AB", "AB CA")
    assert w.shape == (1, 1, (N - (6000 // 256 + 1) - 1) * 256) and w.dtype == np.int16
    assert np.array_equal(w, onnxruntime.synthesize(eng, audio, "ABC AB", "AB CA", vocab, seed=1))   # seeded
    eng.close()


def test_indextts_f_session_like_the_reference_call(tmp_path, golden_dir):
    """ort_session_F.run_with_ort_values([generated_wav], input_feed_F) — Inference_IndexTTS_ONNX.py:787."""
    g = np.load(os.path.join(golden_dir, "indextts_f.npz"))
    cfg = BigVGANConfig.indextts()
    wfile = tmp_path / "ixf.npy"
    np.save(wfile, W.pack_bigvgan(cfg, W.synth_state(W.bigvgan_spec(cfg), 9527)))
    path = onnxruntime.save_model(str(tmp_path / "IndexTTS_F.mi355.json"), "IndexTTS_F", cfg, str(wfile), "f32")
    sess = onnxruntime.InferenceSession(path, sess_options=onnxruntime.SessionOptions(), providers=[])
    in_names = [a.name for a in sess.get_inputs()]
    feed = {f"save_bigvgan_conds_{i}": onnxruntime.OrtValue.ortvalue_from_numpy(g[f"cond{i}"], "cpu", 0) for i in range(6)}
    feed["bigvgan_cond_layer_speaker_embedding"] = onnxruntime.OrtValue.ortvalue_from_numpy(g["cond_pre"], "cpu", 0)
    feed[in_names[-1]] = onnxruntime.OrtValue.ortvalue_from_numpy(g["latent"], "cpu", 0)
    wav = sess.run_with_ort_values([sess.get_outputs()[0].name], feed)[0].numpy()
    assert np.abs(wav.astype(np.int32) - g["wav_i16"].astype(np.int32)).max() <= 3
