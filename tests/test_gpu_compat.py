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


def test_f5_fp16_transformer_export_through_facade(tmp_path, golden_dir):
    """The reference's use_fp16_transformer export (Export_F5.py:20): graphs A / B / C exchange noise, RoPE tables,
    cat_mel_text(_drop) and denoised as float16 (:139-140, :198-199, :348-349), q / k carry the extra x0.1 and the scores the
    x100 (:321-326, fp16/modules.py:467).  The façade's sessions for F5Config(ref_fp16_attn=True) declare and exchange float16,
    refuse float32 feeds like ORT would, and land inside the f16 gate of the reference chain's fp32 waveform."""
    import dataclasses
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    cfg = dataclasses.replace(F5Config.small(), ref_fp16_attn=True)
    wfile = tmp_path / "f5_weights_fp16_export.npy"
    np.save(wfile, W.pack_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527)))
    paths = {k: onnxruntime.save_model(str(tmp_path / f"{k}.mi355.json"), k, cfg, str(wfile), "f16")
             for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")}
    A = onnxruntime.InferenceSession(paths["F5_Preprocess"])
    B = onnxruntime.InferenceSession(paths["F5_Transformer"])
    Cc = onnxruntime.InferenceSession(paths["F5_Decode"])
    assert {a.name: a.type for a in A.get_outputs()}["cat_mel_text"] == "tensor(float16)"
    assert {a.name: a.type for a in B.get_inputs()}["rope_cos_q"] == "tensor(float16)"
    assert B.get_outputs()[0].type == "tensor(float16)" and Cc.get_inputs()[0].type == "tensor(float16)"
    in_A, out_A = [a.name for a in A.get_inputs()], [a.name for a in A.get_outputs()]
    in_B, out_B = [a.name for a in B.get_inputs()], [a.name for a in B.get_outputs()]
    N = int(g["pre_N"])
    noise, cq, sq, ck, sk, cmt, cmtd, rsl = A.run(out_A, {in_A[0]: g["pre_audio"].reshape(1, 1, -1), in_A[1]: g["pre_text_ids"].reshape(1, -1),
                                                          in_A[2]: np.array([N], dtype=np.int64)})
    for a in (noise, cq, sq, ck, sk, cmt, cmtd):
        assert a.dtype == np.float16
    assert np.array_equal(cq.astype(np.float32)[0, 0], g["pre_rope_cos_q"])              # the tables are fp16 values in both exports
    noise = g["dit_noise"][None].astype(np.float16)                                     # Export_F5.py:140: noise.half()
    time_step = np.array([0], dtype=np.int32)
    for i in range(cfg.nfe_step - 1):
        noise, time_step = B.run([out_B[0], out_B[1]], {in_B[0]: noise, in_B[1]: cq, in_B[2]: sq, in_B[3]: ck, in_B[4]: sk,
                                                      in_B[5]: cmt, in_B[6]: cmtd, in_B[7]: time_step})
        assert noise.dtype == np.float16
    wav = Cc.run(None, {Cc.get_inputs()[0].name: noise, Cc.get_inputs()[1].name: rsl})[0]
    assert wav.dtype == np.int16 and wav.shape == (1, 1, g["e2e_i16"].shape[0])
    err = np.sqrt(np.mean(((wav[0, 0].astype(np.float64) - g["e2e_i16"]) / 32767.0) ** 2))
    print(f"fp16-transformer export through the facade (float16 graph I/O, fp16 sampler state between runs): waveform rms {err:.2e}")
    assert err < 5e-4, err                                                              # achieved 5.8e-5
    with pytest.raises(onnxruntime.InvalidArgument):
        B.run(None, {in_B[0]: noise.astype(np.float32), in_B[1]: cq, in_B[2]: sq, in_B[3]: ck, in_B[4]: sk, in_B[5]: cmt, in_B[6]: cmtd,
                     in_B[7]: time_step})


def test_f5_facade_fuse_nfe_seed_and_rope_inputs(tmp_path, golden_dir):
    """FUSE_NFE > 1 (`for i in range(0, NFE_STEP - 1, FUSE_NFE)`, F5-TTS-ONNX-Inference.py:291): the transformer graph advances
    fuse_step Euler steps per run; repeated preprocess runs draw fresh noise like ORT's generator; RoPE feeds that are not
    graph A's tables are refused instead of silently ignored."""
    import dataclasses
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    base = F5Config.small()
    cfg = dataclasses.replace(base, nfe_step=7, fuse_step=2)       # 6 Euler steps = 3 runs of 2
    st = W.synth_state(W.f5_spec(cfg), 9527)
    wfile = tmp_path / "f5_weights.npy"
    np.save(wfile, W.pack_f5(cfg, st))
    pa = onnxruntime.save_model(str(tmp_path / "A.mi355.json"), "F5_Preprocess", cfg, str(wfile), "f32")
    pb = onnxruntime.save_model(str(tmp_path / "B.mi355.json"), "F5_Transformer", cfg, str(wfile), "f32")
    onnxruntime.set_seed(9527)
    A, B = onnxruntime.InferenceSession(pa), onnxruntime.InferenceSession(pb)
    in_A, out_A = [a.name for a in A.get_inputs()], [a.name for a in A.get_outputs()]
    in_B, out_B = [a.name for a in B.get_inputs()], [a.name for a in B.get_outputs()]
    feedA = {in_A[0]: g["pre_audio"].reshape(1, 1, -1), in_A[1]: g["pre_text_ids"].reshape(1, -1),
             in_A[2]: np.array([int(g["pre_N"])], dtype=np.int64)}
    n1, cq, sq, ck, sk, cmt, cmtd, rsl = A.run(out_A, feedA)
    n2 = A.run(out_A, feedA)[0]
    assert not np.array_equal(n1, n2)                           # the generator advanced
    onnxruntime.set_seed(9527)
    assert np.array_equal(A.run(out_A, feedA)[0], n1)           # and is reproducible from the seed
    noise, time_step = g["dit_noise"][None].copy(), np.array([0], dtype=np.int32)
    NFE_STEP, FUSE_NFE, runs = cfg.nfe_step, cfg.fuse_step, 0
    for i in range(0, NFE_STEP - 1, FUSE_NFE):
        noise, time_step = B.run(out_B, {in_B[0]: noise, in_B[1]: cq, in_B[2]: sq, in_B[3]: ck, in_B[4]: sk, in_B[5]: cmt,
                                         in_B[6]: cmtd, in_B[7]: time_step})
        runs += 1
    assert runs == 3 and int(time_step[0]) == 6
    eng = B._eng
    want = eng.sample(g["dit_noise"][None], cmt, cmtd)
    assert np.array_equal(noise, want)                          # 3 fused runs == the 6-step device loop
    with pytest.raises(onnxruntime.InvalidArgument):
        B.run(out_B, {in_B[0]: noise, in_B[1]: cq * 0.5, in_B[2]: sq, in_B[3]: ck, in_B[4]: sk, in_B[5]: cmt, in_B[6]: cmtd,
                      in_B[7]: np.array([0], dtype=np.int32)})


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


def test_indextts_a_session_like_the_reference_call(tmp_path, golden_dir):
    """all_outputs_A = ort_session_A.run_with_ort_values(out_name_A, {in_name_A0: audio}) and the hand-over of its first outputs to
    session F's feed — Inference_IndexTTS_ONNX.py:700-707; output names / order of the export (Export_IndexTTS.py:337-355)."""
    from mi355tts.config import IndexCondConfig
    g = np.load(os.path.join(golden_dir, "indextts_a.npz"))
    cfg = IndexCondConfig.small()
    wfile = tmp_path / "ixa.npy"
    np.save(wfile, W.pack_cond(cfg, W.synth_state(W.cond_spec(cfg), 9527)))
    path = onnxruntime.save_model(str(tmp_path / "IndexTTS_A.mi355.json"), "IndexTTS_A", cfg, str(wfile), "f32")
    ort_session_A = onnxruntime.InferenceSession(path, sess_options=onnxruntime.SessionOptions(), providers=[])
    in_name_A0 = ort_session_A.get_inputs()[0].name
    out_name_A = [o.name for o in ort_session_A.get_outputs()]
    assert in_name_A0 == "audio" and out_name_A == ["save_bigvgan_conds_0", "save_bigvgan_conds_1", "bigvgan_cond_layer_speaker_embedding",
                                                    "conds_latent"]
    audio = onnxruntime.OrtValue.ortvalue_from_numpy(g["r_audio"].reshape(1, 1, -1), "cpu", 0)
    all_outputs_A = ort_session_A.run_with_ort_values(out_name_A, {in_name_A0: audio})
    for i in range(2):
        o = all_outputs_A[i].numpy()
        assert o.shape == (1, cfg.voc_channels[i], 1)
        np.testing.assert_allclose(o.reshape(-1), g[f"r_cond_{i}"], atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(all_outputs_A[2].numpy().reshape(-1), g["r_cond_layer"], atol=2e-4, rtol=1e-3)
    lat = all_outputs_A[3].numpy()
    assert lat.shape == (1, cfg.latents, cfg.model_dim)
    np.testing.assert_allclose(lat[0], g["r_conds_latent"], atol=5e-4, rtol=1e-3)
    with pytest.raises(onnxruntime.InvalidArgument):
        ort_session_A.run(out_name_A, {in_name_A0: g["r_audio"].astype(np.float32).reshape(1, 1, -1)})


@pytest.mark.parametrize("device_type", ["cpu", "cuda"])
def test_indextts_gpt_driver_loop_through_facade(tmp_path, golden_dir, device_type):
    """Inference_IndexTTS_ONNX.py:619-800 for one sentence with sessions B, C, D, E: same variable names and feed
    bookkeeping as the reference driver; out_key/out_value OrtValues are fed straight back as in_key/in_value.
    device_type 'cuda' (round 5): every `ortvalue_from_numpy(x, device_type, DEVICE_ID)` of the driver is a device-resident value
    (the sessions of this model read them back through `.numpy()`: same tokens, same hidden states)."""
    from mi355tts.config import IndexGPTConfig
    g = np.load(os.path.join(golden_dir, "indextts_gpt.npz"))
    cfg = IndexGPTConfig.small()
    wfile = tmp_path / "gpt_weights.npy"
    np.save(wfile, W.pack_gpt(cfg, W.synth_state(W.gpt_spec(cfg), 9527)))
    paths = {k: onnxruntime.save_model(str(tmp_path / f"{k}.mi355.json"), k, cfg, str(wfile), "f32")
             for k in ("IndexTTS_B", "IndexTTS_C", "IndexTTS_D", "IndexTTS_E")}
    DEVICE_ID = 0
    REPEAT_PENALITY, PENALITY_RANGE = float(g["gen_params"][0]), int(g["gen_params"][1])
    STOP_TOKEN = [cfg.stop_mel_token]
    ort_session_B = onnxruntime.InferenceSession(paths["IndexTTS_B"])
    ort_session_C = onnxruntime.InferenceSession(paths["IndexTTS_C"])
    ort_session_D = onnxruntime.InferenceSession(paths["IndexTTS_D"])
    ort_session_E = onnxruntime.InferenceSession(paths["IndexTTS_E"])
    in_name_B0, out_name_B0 = ort_session_B.get_inputs()[0].name, ort_session_B.get_outputs()[0].name
    in_name_C = [a.name for a in ort_session_C.get_inputs()]
    out_name_C = [a.name for a in ort_session_C.get_outputs()]
    in_name_D = [a.name for a in ort_session_D.get_inputs()]
    out_name_D = [a.name for a in ort_session_D.get_outputs()]
    model_E_dtype = np.float16 if "float16" in ort_session_E._inputs_meta[0].type else np.float32
    in_names_E = [a.name for a in ort_session_E.get_inputs()]
    out_name_E = [a.name for a in ort_session_E.get_outputs()]
    amount_of_outputs_E = len(out_name_E)
    num_layers = (amount_of_outputs_E - 3) // 2
    assert num_layers == cfg.layers and len(in_names_E) == 2 * num_layers + 5
    num_layers_2 = num_layers * 2
    last_input_indices_E, last_output_indices_E = len(in_names_E) - 1, amount_of_outputs_E - 1
    second_last_output_indices_E = amount_of_outputs_E - 2
    OV = onnxruntime.OrtValue.ortvalue_from_numpy
    init_gpt_ids = OV(np.array([[cfg.start_mel_token]], dtype=np.int32), device_type, DEVICE_ID)
    init_gen_len = OV(np.array([0], dtype=np.int64), device_type, DEVICE_ID)
    init_ids_len_1 = OV(np.array([1], dtype=np.int64), device_type, DEVICE_ID)
    init_history_len = OV(np.array([0], dtype=np.int64), device_type, DEVICE_ID)
    init_attention_mask_0 = OV(np.array([0], dtype=np.int8), device_type, DEVICE_ID)
    init_attention_mask_1 = OV(np.array([1], dtype=np.int8), device_type, DEVICE_ID)
    m = ort_session_E._inputs_meta
    init_past_keys_E = OV(np.zeros((m[0].shape[0], m[0].shape[1], 0), dtype=model_E_dtype), device_type, DEVICE_ID)
    init_past_values_E = OV(np.zeros((m[num_layers].shape[0], 0, m[num_layers].shape[2]), dtype=model_E_dtype), device_type, DEVICE_ID)
    repeat_penality = OV(np.ones((1, m[num_layers_2 + 1].shape[1]), dtype=model_E_dtype), device_type, DEVICE_ID)
    input_feed_E = {in_names_E[last_input_indices_E]: init_attention_mask_1, in_names_E[num_layers_2]: init_history_len,
                    in_names_E[num_layers_2 + 1]: repeat_penality}
    for i in range(num_layers):
        input_feed_E[in_names_E[i]] = init_past_keys_E
    for i in range(num_layers, num_layers_2):
        input_feed_E[in_names_E[i]] = init_past_values_E

    conds_latent = OV(g["conds_latent"], device_type, DEVICE_ID)
    text_ids = OV(g["text_ids"], device_type, DEVICE_ID)
    text_hidden_state = ort_session_B.run_with_ort_values([out_name_B0], {in_name_B0: text_ids})[0]
    gpt_hidden_state, gen_len = ort_session_C.run_with_ort_values(out_name_C, {in_name_C[0]: init_gpt_ids, in_name_C[1]: init_gen_len})
    gpt_hidden_state, concat_len = ort_session_D.run_with_ort_values(
        out_name_D, {in_name_D[0]: conds_latent, in_name_D[1]: text_hidden_state, in_name_D[2]: gpt_hidden_state})
    np.testing.assert_allclose(gpt_hidden_state.numpy(), g["D_hidden"], atol=1e-6, rtol=0)
    generate_limit = 13 + len(g["gen_tokens"]) - onnxruntime.OrtValue.numpy(concat_len)
    input_feed_E[in_names_E[num_layers_2 + 2]] = concat_len
    save_last_hidden_state, save_max_logits_ids = [], []
    reset_penality = num_decode = 0
    while num_decode < generate_limit:
        input_feed_E[in_names_E[num_layers_2 + 3]] = gpt_hidden_state
        all_outputs_E = ort_session_E.run_with_ort_values(out_name_E, input_feed_E)
        max_logit_ids = onnxruntime.OrtValue.numpy(all_outputs_E[last_output_indices_E])
        save_max_logits_ids.append(max_logit_ids)
        save_last_hidden_state.append(all_outputs_E[second_last_output_indices_E])
        num_decode += 1
        if max_logit_ids in STOP_TOKEN:
            break
        if num_decode < 2:
            input_feed_E[in_names_E[last_input_indices_E]] = init_attention_mask_0
            input_feed_E[in_names_E[num_layers_2 + 2]] = init_ids_len_1
        for i in range(second_last_output_indices_E):
            input_feed_E[in_names_E[i]] = all_outputs_E[i]
        repeat_penality = onnxruntime.OrtValue.numpy(repeat_penality)
        repeat_penality[:, max_logit_ids] = REPEAT_PENALITY
        if (num_decode > PENALITY_RANGE) and (save_max_logits_ids[reset_penality] != max_logit_ids):
            repeat_penality[:, save_max_logits_ids[reset_penality]] = 1.0
            reset_penality += 1
        repeat_penality = OV(repeat_penality, device_type, DEVICE_ID)
        input_feed_E[in_names_E[num_layers_2 + 1]] = repeat_penality
        gpt_hidden_state, gen_len = ort_session_C.run_with_ort_values(
            out_name_C, {in_name_C[0]: all_outputs_E[last_output_indices_E], in_name_C[1]: gen_len})
    toks = [int(t.reshape(-1)[0]) for t in save_max_logits_ids]
    assert toks == [int(x) for x in g["gen_tokens"]]
    hid = np.concatenate([onnxruntime.OrtValue.numpy(h) for h in save_last_hidden_state], axis=0)
    np.testing.assert_allclose(hid, g["gen_hidden"], atol=3e-4, rtol=0)
    # the final cache can still be materialised in the reference's layouts
    np.testing.assert_allclose(all_outputs_E[0].numpy(), g["gen_key0"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(all_outputs_E[num_layers + 1].numpy(), g["gen_value1"], atol=3e-4, rtol=0)
    # next sentence: the empty caches are fed again (:796-800) -> history restarts
    input_feed_E[in_names_E[last_input_indices_E]] = init_attention_mask_1
    input_feed_E[in_names_E[num_layers_2]] = init_history_len
    for i in range(num_layers):
        input_feed_E[in_names_E[i]] = init_past_keys_E
    for i in range(num_layers, num_layers_2):
        input_feed_E[in_names_E[i]] = init_past_values_E
    input_feed_E[in_names_E[num_layers_2 + 2]] = concat_len
    input_feed_E[in_names_E[num_layers_2 + 3]] = OV(g["D_hidden"], device_type, DEVICE_ID)
    input_feed_E[in_names_E[num_layers_2 + 1]] = OV(np.ones((1, cfg.mel_codes), np.float32), device_type, DEVICE_ID)
    again = ort_session_E.run_with_ort_values(out_name_E, input_feed_E)
    assert int(again[last_output_indices_E].numpy().reshape(-1)[0]) == int(g["gen_tokens"][0])
    assert int(again[num_layers_2].numpy()[0]) == 13
    # stale references are refused, numpy round trips work (plain .run materialises the cache)
    with pytest.raises(onnxruntime.Fail):
        all_outputs_E[0].numpy()
    feed_np = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in input_feed_E.items()}
    outs = ort_session_E.run(None, feed_np)
    assert outs[0].shape == (cfg.heads, cfg.head_dim, 13) and int(outs[-1].reshape(-1)[0]) == int(g["gen_tokens"][0])
    # a history given as arrays (not references) is loaded into the cache: one more step from the golden mid-state
    feed_np = {f"in_key_{i}": g["S_keys_in"][i] for i in range(num_layers)}
    feed_np.update({f"in_value_{i}": g["S_values_in"][i] for i in range(num_layers)})
    feed_np.update({"history_len": np.array([g["S_keys_in"].shape[3]], np.int64), "repeat_penality": g["S_pen"],
                    "ids_len": np.array([1], np.int64), "hidden_state": g["S_hidden_in"],
                    "attention_mask": np.array([0], np.int8)})
    kv, last, tok = ort_session_E.run(["kv_seq_len", "last_hidden_state", "max_logit_id"], feed_np)
    assert int(kv[0]) == g["S_keys_in"].shape[3] + 1 and int(tok[0, 0]) == int(g["S_token"][0, 0])
    np.testing.assert_allclose(last, g["S_last_hidden"], atol=3e-4, rtol=0)
    with pytest.raises(onnxruntime.InvalidArgument):
        ort_session_E.run(None, {**feed_np, "ids_len": np.array([2], np.int64)})


def test_handles_are_thread_safe(golden_dir):
    """SURVEY.md §8b threading: calls on one handle are serialised, different handles run concurrently; ctypes drops
    the GIL for the duration of a call."""
    import threading
    from mi355tts.bigvgan import BigVGANVocoder
    cfg = BigVGANConfig.small()
    st = W.synth_state(W.bigvgan_spec(cfg), 9527)
    a, b = BigVGANVocoder(cfg, st, dtype="f32"), BigVGANVocoder(cfg, st, dtype="f32")
    mels = [W.synth_normal(50 + i, "mel", (1, cfg.num_mels, 40 + 8 * i), std=1.0) for i in range(6)]
    want = [a.run(m) for m in mels]
    got, errs = {}, []

    def work(tid, eng):
        try:
            for rep in range(4):
                for i, m in enumerate(mels):
                    got[(tid, rep, i)] = eng.run(m)
        except Exception as e:                      # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(t, a if t < 3 else b)) for t in range(5)]     # 3 threads share handle a
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for (tid, rep, i), y in got.items():
        np.testing.assert_array_equal(y, want[i])
    a.close(); b.close()


def test_set_option_while_other_threads_run_calls():
    """include/mi355tts.h: "different handles may be used concurrently from different threads" — with mi_set_option in the picture
    (VERDICT r4 #10).  Every entry point holds the shared side of one reader-writer lock, mi_set_option the exclusive side: a thread
    that flips a dispatch option (here `bigvgan_streams`, whose three settings are bit-identical by construction) while two other
    threads run forwards on two handles must neither corrupt a result nor dead-lock, and every flip must succeed."""
    import threading
    from mi355tts import _lib
    from mi355tts.bigvgan import BigVGANVocoder
    cfg = BigVGANConfig.small()
    st = W.synth_state(W.bigvgan_spec(cfg), 9527)
    engs = [BigVGANVocoder(cfg, st, dtype="f32") for _ in range(2)]
    mels = [W.synth_normal(70 + i, "mel", (1, cfg.num_mels, 64 + 16 * i), std=1.0) for i in range(4)]
    want = [engs[0].run(m) for m in mels]
    errs, done, flips = [], threading.Event(), [0]

    def work(eng):
        try:
            for rep in range(25):
                for i, m in enumerate(mels):
                    if not np.array_equal(eng.run(m), want[i]):
                        errs.append(("mismatch", rep, i))
        except Exception as e:                      # pragma: no cover
            errs.append(e)

    def flip():
        try:
            while not done.is_set():
                for v in (1, 2, 3):
                    _lib.set_option("bigvgan_streams", v)
                    flips[0] += 1
        except Exception as e:                      # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    tf = threading.Thread(target=flip)
    [t.start() for t in th]; tf.start()
    [t.join(300) for t in th]
    done.set(); tf.join(60)
    _lib.set_option("bigvgan_streams", 3)
    assert not any(t.is_alive() for t in th) and not tf.is_alive(), "dead-lock"
    assert not errs, errs[:3]
    assert flips[0] >= 3
    for e in engs:
        e.close()


def test_example_script_runs(tmp_path):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from mi355tts import audio_io
    prompt = str(tmp_path / "prompt.wav")
    t = np.arange(44100)
    audio_io.write_wavex(prompt, (8000 * np.sin(2 * np.pi * 330 * t / 44100)).astype(np.int16), 44100)
    out = str(tmp_path / "gen.wav")
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "f5_tts_infer.py"), "--small", "--dtype", "f32",
                        "--prompt", prompt, "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    wav, rate = audio_io.read_wav(out)
    assert rate == 24000 and wav.shape[1] == 1 and wav.shape[0] > 1000 and "RTF" in r.stdout


@pytest.mark.parametrize("device_type", ["cpu", "cuda"])
def test_indextts_example_script_runs(tmp_path, device_type):
    """examples/indextts_infer.py: Inference_IndexTTS_ONNX.py:578-805 end to end (all six sessions, sentence loop, WAVEX write) on
    the reduced models; the host and the device-resident OrtValue forms of the driver must write the same file."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from mi355tts import audio_io
    prompt = str(tmp_path / "prompt.wav")
    t = np.arange(30000)
    audio_io.write_wavex(prompt, (8000 * np.sin(2 * np.pi * 330 * t / 24000)).astype(np.int16), 24000)
    out = str(tmp_path / f"gen_{device_type}.wav")
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "indextts_infer.py"), "--small", "--dtype", "f32", "--prompt", prompt,
                        "--text", "hello there. how are you today?", "--out", out, "--device-type", device_type, "--ignore-stop",
                        "--max-generate-length", "70"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    wav, rate = audio_io.read_wav(out)
    assert rate == 24000 and wav.shape[1] == 1 and wav.shape[0] > 4800 and "RTF" in r.stdout and "Decode Speed" in r.stdout
    ref = tmp_path / "gen_cpu.wav"
    if device_type == "cuda" and ref.exists():
        assert np.array_equal(audio_io.read_wav(str(ref))[0], wav)


def test_bigvgan_example_script_runs(tmp_path):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from mi355tts import audio_io
    outs = []
    for device_type in ("cpu", "cuda"):
        out = str(tmp_path / f"bv_{device_type}.wav")
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "bigvgan_infer.py"), "--small", "--dtype", "f32", "--out", out,
                            "--device-type", device_type], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "RTF" in r.stdout, r.stderr[-2000:]
        outs.append(audio_io.read_wav(out)[0])
    assert outs[0].shape[0] == 512 * 8 + 30 and np.array_equal(outs[0], outs[1])


def _f5_sessions(tmp_path, cfg, dtype="f32"):
    wfile = tmp_path / "f5_weights.npy"
    np.save(wfile, W.pack_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527)))
    paths = {k: onnxruntime.save_model(str(tmp_path / f"{k}.mi355.json"), k, cfg, str(wfile), dtype)
             for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")}
    return [onnxruntime.InferenceSession(paths[k]) for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")]


def test_f5_driver_io_binding_branch_device_resident(tmp_path, golden_dir):
    """The reference's `if device_type:` branch (F5-TTS-ONNX-Inference.py:256-288), line for line: graph A's outputs become
    DEVICE OrtValues (`ortvalue_from_numpy(x, 'cuda', DEVICE_ID)`), outputs 0 / 1 of graph B are bound onto inputs 0 / 7, the
    loop is `run_with_iobinding`, the result comes back with `OrtValue.numpy(io_binding.get_outputs()[0])`.  The façade keeps
    those values in HBM and hands device pointers to the C-ABI (round 5; the round-4 OrtValue was a host array whatever the
    device type).  Against the fixture AND bit-equal to the host-numpy loop."""
    import torch
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    cfg = F5Config.small()
    ort_session_A, ort_session_B, ort_session_C = _f5_sessions(tmp_path, cfg)
    in_A, out_A = [a.name for a in ort_session_A.get_inputs()], [a.name for a in ort_session_A.get_outputs()]
    in_name_B, out_name_B = ort_session_B.get_inputs(), ort_session_B.get_outputs()
    NFE_STEP, FUSE_NFE, DEVICE_ID, device_type = cfg.nfe_step, 1, 0, "cuda"
    audio = g["pre_audio"].reshape(1, 1, -1)
    text_ids = g["pre_text_ids"].reshape(1, -1)
    max_duration = np.array([int(g["pre_N"])], dtype=np.int64)
    time_step = np.array([0], dtype=np.int32)
    noise, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, cat_mel_text, cat_mel_text_drop, ref_signal_len = ort_session_A.run(
        out_A, {in_A[0]: audio, in_A[1]: text_ids, in_A[2]: max_duration})
    noise = g["dit_noise"][None].copy()                          # the fixture's noise from here on
    host_noise, host_ts = noise.copy(), time_step.copy()
    inputs = [
        onnxruntime.OrtValue.ortvalue_from_numpy(noise, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(rope_cos_q, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(rope_sin_q, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(rope_cos_k, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(rope_sin_k, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(cat_mel_text, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(cat_mel_text_drop, device_type, DEVICE_ID),
        onnxruntime.OrtValue.ortvalue_from_numpy(time_step, device_type, DEVICE_ID)
    ]
    assert all(v.is_device() and v.device_name() == "cuda" for v in inputs)
    assert inputs[1].shape() == list(rope_cos_q.shape) and inputs[3].shape() == list(rope_cos_k.shape)
    assert inputs[1]._t.untyped_storage().nbytes() == rope_cos_q.shape[2] * rope_cos_q.shape[3] * 4     # the broadcast axes did not cross PCIe
    outputs = [inputs[0], inputs[-1]]
    io_binding = ort_session_B.io_binding()
    for i in range(len(inputs)):
        io_binding.bind_ortvalue_input(name=in_name_B[i].name, ortvalue=inputs[i])
    for i in range(len(outputs)):
        io_binding.bind_ortvalue_output(name=out_name_B[i].name, ortvalue=outputs[i])
    ptr0 = inputs[0].data_ptr()
    for i in range(0, NFE_STEP - 1, FUSE_NFE):
        ort_session_B.run_with_iobinding(io_binding)
    assert io_binding.get_outputs()[0] is inputs[0] and inputs[0].data_ptr() == ptr0          # advanced in place, as bound
    assert int(onnxruntime.OrtValue.numpy(io_binding.get_outputs()[1])[0]) == NFE_STEP - 1
    assert int(inputs[-1]._t.cpu()[0]) == NFE_STEP - 1                                       # the device value itself, not just the mirror
    noise_dev = onnxruntime.OrtValue.numpy(io_binding.get_outputs()[0])
    # the `else:` branch on the same inputs (host numpy through run): same engine, same steps -> the same bits
    for i in range(0, NFE_STEP - 1, FUSE_NFE):
        host_noise, host_ts = ort_session_B.run([out_name_B[0].name, out_name_B[1].name], {
            in_name_B[0].name: host_noise, in_name_B[1].name: rope_cos_q, in_name_B[2].name: rope_sin_q, in_name_B[3].name: rope_cos_k,
            in_name_B[4].name: rope_sin_k, in_name_B[5].name: cat_mel_text, in_name_B[6].name: cat_mel_text_drop, in_name_B[7].name: host_ts})
    assert np.array_equal(noise_dev, host_noise)
    generated_signal = ort_session_C.run([ort_session_C.get_outputs()[0].name], {
        ort_session_C.get_inputs()[0].name: noise_dev, ort_session_C.get_inputs()[1].name: ref_signal_len})[0]
    err = np.sqrt(np.mean(((generated_signal[0, 0].astype(np.float64) - g["e2e_i16"]) / 32767.0) ** 2))
    assert err < 5e-4, err
    # run_with_ort_values on device values without bound outputs: the inputs stay untouched, the outputs are new device values
    fresh = onnxruntime.OrtValue.ortvalue_from_numpy(g["dit_noise"][None].copy(), device_type, DEVICE_ID)
    ts0 = onnxruntime.OrtValue.ortvalue_from_numpy(np.array([0], dtype=np.int32), device_type, DEVICE_ID)
    feed = {in_name_B[i].name: inputs[i] for i in range(1, 7)}
    feed[in_name_B[0].name], feed[in_name_B[7].name] = fresh, ts0
    den, ts1 = ort_session_B.run_with_ort_values([o.name for o in out_name_B], feed)
    assert den.is_device() and den is not fresh and np.array_equal(fresh.numpy(), g["dit_noise"][None])
    assert int(ts1.numpy()[0]) == 1 and int(ts0.numpy()[0]) == 0
    one_host, _ = ort_session_B.run([o.name for o in out_name_B], {
        in_name_B[0].name: g["dit_noise"][None].copy(), in_name_B[1].name: rope_cos_q, in_name_B[2].name: rope_sin_q,
        in_name_B[3].name: rope_cos_k, in_name_B[4].name: rope_sin_k, in_name_B[5].name: cat_mel_text,
        in_name_B[6].name: cat_mel_text_drop, in_name_B[7].name: np.array([0], dtype=np.int32)})
    assert np.array_equal(den.numpy(), one_host)
    # a RoPE value that is not graph A's table is refused on the device path too; wrong dtype as well
    bad = dict(feed)
    bad[in_name_B[1].name] = onnxruntime.OrtValue.ortvalue_from_numpy(np.ascontiguousarray(rope_cos_q) * 0.5, device_type, DEVICE_ID)
    with pytest.raises(onnxruntime.InvalidArgument):
        ort_session_B.run_with_ort_values(None, bad)
    bad = dict(feed)
    bad[in_name_B[0].name] = onnxruntime.OrtValue._from_tensor(fresh._t.double())
    with pytest.raises(onnxruntime.InvalidArgument):
        ort_session_B.run_with_ort_values(None, bad)
    del torch


def test_bigvgan_run_with_ort_values_on_a_device_value(tmp_path, golden_dir):
    """Export_BigVGAN.py:160-170 with device_type = 'cuda': the dummy mel is a device OrtValue, the waveform comes back as one."""
    g = np.load(os.path.join(golden_dir, "bigvgan_small.npz"))
    cfg = BigVGANConfig.small()
    wfile = tmp_path / "bv.npy"
    np.save(wfile, W.pack_bigvgan(cfg, W.synth_state(W.bigvgan_spec(cfg), 9527)))
    path = onnxruntime.save_model(str(tmp_path / "BigVGAN.mi355.json"), "BigVGAN", cfg, str(wfile), "f32")
    ort_session_A = onnxruntime.InferenceSession(path, sess_options=onnxruntime.SessionOptions(), providers=[], provider_options=None)
    test_dummy = onnxruntime.OrtValue.ortvalue_from_numpy(np.ones((1, ort_session_A._inputs_meta[0].shape[1], 12), dtype=np.float32), "cuda", 0)
    output = ort_session_A.run_with_ort_values([ort_session_A.get_outputs()[0].name], {ort_session_A.get_inputs()[0].name: test_dummy})
    assert output[0].is_device() and output[0].data_type() == "tensor(int16)"
    w = onnxruntime.OrtValue.numpy(output[0])
    assert np.abs(w.astype(np.int32) - g["gen_i16_ones"].astype(np.int32)).max() <= 1
    host = ort_session_A.run_with_ort_values([ort_session_A.get_outputs()[0].name], {
        ort_session_A.get_inputs()[0].name: onnxruntime.OrtValue.ortvalue_from_numpy(np.ones((1, cfg.num_mels, 12), dtype=np.float32), "cpu", 0)})
    assert np.array_equal(host[0].numpy(), w)
