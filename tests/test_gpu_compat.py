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
    # the same export through io-binding on DEVICE values: outputs 0 / 1 bound onto inputs 0 / 7 must ADVANCE them —
    # a session that hands back fresh values instead would recompute step 0 thirty-one times
    dev = [onnxruntime.OrtValue.ortvalue_from_numpy(x, "cuda", 0) for x in
           (g["dit_noise"][None].astype(np.float16), cq, sq, ck, sk, cmt, cmtd, np.array([0], dtype=np.int32))]
    binding = B.io_binding()
    for name, v in zip(in_B, dev):
        binding.bind_ortvalue_input(name=name, ortvalue=v)
    binding.bind_ortvalue_output(name=out_B[0], ortvalue=dev[0])
    binding.bind_ortvalue_output(name=out_B[1], ortvalue=dev[7])
    for i in range(cfg.nfe_step - 1):
        B.run_with_iobinding(binding)
    assert binding.get_outputs()[0] is dev[0] and dev[0].is_device() and dev[0].data_type() == "tensor(float16)"
    assert int(dev[7].numpy()[0]) == cfg.nfe_step - 1
    assert np.array_equal(dev[0].numpy(), noise)                                        # the host loop above, bit for bit


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


# ---- IndexTTS graphs B / C / D / E behind the façade: the CONTRACT a driver relies on, as a table the test walks ------------------
# (name, ONNX element type, rank) per input / output, in export order (Export_IndexTTS.py:203-297: input_names / output_names of the
# four torch.onnx.export calls).  `{i}` rows are repeated per GPT layer.
_GPT_IO = {
    "IndexTTS_B": ([("text_ids", "int32", 2)], [("text_hidden_state", "float", 3)]),
    "IndexTTS_C": ([("gpt_ids", "int32", 2), ("kv_seq_len", "int64", 1)], [("gpt_hidden_state", "float", 3), ("next_kv_seq_len", "int64", 1)]),
    "IndexTTS_D": ([("embed_x", "float", 3), ("embed_y", "float", 3), ("embed_z", "float", 3)],
                   [("concat_hidden_state", "float", 3), ("concat_len", "int64", 1)]),
    "IndexTTS_E": ([("in_key_{i}", "float", 3), ("in_value_{i}", "float", 3), ("history_len", "int64", 1), ("repeat_penality", "float", 2),
                    ("ids_len", "int64", 1), ("hidden_state", "float", 3), ("attention_mask", "int8", 1)],
                   [("out_key_{i}", "float", 3), ("out_value_{i}", "float", 3), ("kv_seq_len", "int64", 1), ("last_hidden_state", "float", 2),
                    ("max_logit_id", "int32", 2)]),
}


def _expand_io(rows, layers):
    out = []
    for name, ty, rank in rows:
        out += [(name.format(i=i), ty, rank) for i in range(layers)] if "{i}" in name else [(name, ty, rank)]
    return out


class _GreedyMelDecoder:
    """What a caller of the four sessions has to do for one sentence — stated against the contract, by NAME: prompt = D(conds, B(text),
    C(start)); then E on the prompt with attention_mask 1 and an empty history, and E on one row at a time with attention_mask 0, the
    out_key / out_value VALUES of a step being the in_key / in_value of the next (references, never copied); the token of a step
    goes through C to become the next row.  The repeat penalty is host state: a chosen code's entry is set to `value`, and once more
    than `window` codes have been chosen the oldest chosen code is released back to 1 unless it is the code just chosen."""

    def __init__(self, sessions, place, layers, mel_codes, value, window):
        self.B, self.C, self.D, self.E = sessions
        self.place, self.layers, self.value, self.window = place, layers, value, window
        self.penalty = np.ones((1, mel_codes), np.float32)
        self.keys_in = [f"in_key_{i}" for i in range(layers)]
        self.vals_in = [f"in_value_{i}" for i in range(layers)]
        self.e_outs = [o.name for o in self.E.get_outputs()]
        self.c_outs = [o.name for o in self.C.get_outputs()]
        meta = {a.name: a for a in self.E.get_inputs()}
        k, v = meta["in_key_0"].shape, meta["in_value_0"].shape
        self.empty_k = place(np.zeros((k[0], k[1], 0), np.float32))
        self.empty_v = place(np.zeros((v[0], 0, v[2]), np.float32))

    def prompt(self, conds_latent, text_ids, start_code):
        text_h = self.B.run_with_ort_values(["text_hidden_state"], {"text_ids": self.place(text_ids)})[0]
        row, self.position = self.C.run_with_ort_values(self.c_outs, {"gpt_ids": self.place(np.array([[start_code]], np.int32)),
                                                                      "kv_seq_len": self.place(np.array([0], np.int64))})
        return self.D.run_with_ort_values([o.name for o in self.D.get_outputs()],
                                          {"embed_x": self.place(conds_latent), "embed_y": text_h, "embed_z": row})

    def fresh_feed(self, rows, n_rows):
        feed = {k: self.empty_k for k in self.keys_in}
        feed.update({v: self.empty_v for v in self.vals_in})
        feed.update(history_len=self.place(np.array([0], np.int64)), repeat_penality=self.place(self.penalty.copy()), ids_len=n_rows,
                    hidden_state=rows, attention_mask=self.place(np.array([1], np.int8)))
        return feed

    def decode(self, rows, n_rows, limit, stop_codes):
        feed = self.fresh_feed(rows, n_rows)
        chosen, hidden, released = [], [], 0
        out = None
        while len(chosen) < limit:
            out = dict(zip(self.e_outs, self.E.run_with_ort_values(self.e_outs, feed)))
            code = int(out["max_logit_id"].numpy().reshape(-1)[0])
            chosen.append(code)
            hidden.append(out["last_hidden_state"])
            if code in stop_codes:
                break
            self.penalty[0, code] = self.value
            if len(chosen) > self.window and chosen[released] != code:
                self.penalty[0, chosen[released]] = 1.0
                released += 1
            row, self.position = self.C.run_with_ort_values(self.c_outs, {"gpt_ids": out["max_logit_id"], "kv_seq_len": self.position})
            feed = {f"in_key_{i}": out[f"out_key_{i}"] for i in range(self.layers)}
            feed.update({f"in_value_{i}": out[f"out_value_{i}"] for i in range(self.layers)})
            feed.update(history_len=out["kv_seq_len"], repeat_penality=self.place(self.penalty.copy()),
                        ids_len=self.place(np.array([1], np.int64)), hidden_state=row, attention_mask=self.place(np.array([0], np.int8)))
        return chosen, hidden, out


@pytest.mark.parametrize("device_type", ["cpu", "cuda"])
def test_indextts_gpt_sessions_contract(tmp_path, golden_dir, device_type):
    """Sessions B, C, D, E of IndexTTS as a drop-in for the reference driver's sentence loop (Inference_IndexTTS_ONNX.py:723-800):
    names / element types / ranks of every input and output, values passed as OrtValues on the host or on the device, cache values
    consumed by reference, the tokens and hidden states of the reference fixture, cache restart, stale references, the numpy form."""
    from mi355tts.config import IndexGPTConfig
    g = np.load(os.path.join(golden_dir, "indextts_gpt.npz"))
    cfg = IndexGPTConfig.small()
    wfile = tmp_path / "gpt_weights.npy"
    np.save(wfile, W.pack_gpt(cfg, W.synth_state(W.gpt_spec(cfg), 9527)))
    sess = {}
    for graph, (ins, outs) in _GPT_IO.items():
        sess[graph] = onnxruntime.InferenceSession(onnxruntime.save_model(str(tmp_path / f"{graph}.mi355.json"), graph, cfg, str(wfile), "f32"))
        for have, want in ((sess[graph].get_inputs(), _expand_io(ins, cfg.layers)), (sess[graph].get_outputs(), _expand_io(outs, cfg.layers))):
            assert [(a.name, a.type, len(a.shape)) for a in have] == [(n, f"tensor({t})", r) for n, t, r in want], graph
    place = lambda x: onnxruntime.OrtValue.ortvalue_from_numpy(np.ascontiguousarray(x), device_type, 0)
    dec = _GreedyMelDecoder([sess[k] for k in _GPT_IO], place, cfg.layers, cfg.mel_codes, float(g["gen_params"][0]), int(g["gen_params"][1]))
    rows, n_rows = dec.prompt(g["conds_latent"], g["text_ids"], cfg.start_mel_token)
    assert int(n_rows.numpy()[0]) == 13
    np.testing.assert_allclose(rows.numpy(), g["D_hidden"], atol=1e-6, rtol=0)
    want_tokens = [int(x) for x in g["gen_tokens"]]
    chosen, hidden, last = dec.decode(rows, n_rows, len(want_tokens), [cfg.stop_mel_token])
    assert chosen == want_tokens
    np.testing.assert_allclose(np.concatenate([h.numpy() for h in hidden], axis=0), g["gen_hidden"], atol=3e-4, rtol=0)
    # the cache behind the last step's references, in the reference's layouts
    np.testing.assert_allclose(last["out_key_0"].numpy(), g["gen_key0"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(last["out_value_1"].numpy(), g["gen_value1"], atol=3e-4, rtol=0)
    # a new sentence = empty caches fed again: the history restarts, the first token is the fixture's first token
    dec.penalty[:] = 1.0
    again = dict(zip(dec.e_outs, sess["IndexTTS_E"].run_with_ort_values(dec.e_outs, dec.fresh_feed(place(g["D_hidden"]), n_rows))))
    assert int(again["max_logit_id"].numpy().reshape(-1)[0]) == want_tokens[0] and int(again["kv_seq_len"].numpy()[0]) == 13
    with pytest.raises(onnxruntime.Fail):                     # ... and the references of the sentence before are stale now
        last["out_key_0"].numpy()
    # plain run() on arrays: the cache is materialised for the caller
    arrays = {k: v.numpy() for k, v in dec.fresh_feed(place(g["D_hidden"]), n_rows).items()}
    outs = sess["IndexTTS_E"].run(None, arrays)
    assert outs[0].shape == (cfg.heads, cfg.head_dim, 13) and int(outs[-1].reshape(-1)[0]) == want_tokens[0]
    # a history handed over as ARRAYS is loaded into the cache: one step from the fixture's mid-sentence state
    mid = {f"in_key_{i}": g["S_keys_in"][i] for i in range(cfg.layers)}
    mid.update({f"in_value_{i}": g["S_values_in"][i] for i in range(cfg.layers)})
    mid.update(history_len=np.array([g["S_keys_in"].shape[3]], np.int64), repeat_penality=g["S_pen"], ids_len=np.array([1], np.int64),
               hidden_state=g["S_hidden_in"], attention_mask=np.array([0], np.int8))
    kv, hid, tok = sess["IndexTTS_E"].run(["kv_seq_len", "last_hidden_state", "max_logit_id"], mid)
    assert int(kv[0]) == g["S_keys_in"].shape[3] + 1 and int(tok[0, 0]) == int(g["S_token"][0, 0])
    np.testing.assert_allclose(hid, g["S_last_hidden"], atol=3e-4, rtol=0)
    with pytest.raises(onnxruntime.InvalidArgument):          # ids_len must say what hidden_state holds
        sess["IndexTTS_E"].run(None, {**mid, "ids_len": np.array([2], np.int64)})


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


# graph B of F5 (F5_Transformer): which output may be bound onto which input — the aliasing the reference's io-binding loop
# relies on (F5-TTS-ONNX-Inference.py:256-288): output k advances input _F5_B_ALIAS[k] in place
_F5_B_INPUTS = ("noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text", "cat_mel_text_drop", "time_step")
_F5_B_ALIAS = {"denoised": "noise", "time_step": "time_step"}


def test_f5_driver_io_binding_branch_device_resident(tmp_path, golden_dir):
    """Graph B's io-binding contract with DEVICE-resident values: every graph-A output placed on the device, the two outputs bound
    onto the inputs they advance, `run_with_iobinding` NFE - 1 times, the result read back from the binding.  Checked: the values
    are device values (broadcast RoPE axes not expanded), the bound buffers advance IN PLACE (same pointer, device copy itself), the
    result equals the host-array loop bit for bit and the reference fixture within its gate; unbound `run_with_ort_values` leaves its
    inputs alone; foreign RoPE tables and wrong dtypes are refused."""
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    cfg = F5Config.small()
    pre, dit, voc = _f5_sessions(tmp_path, cfg)
    assert tuple(a.name for a in dit.get_inputs()) == _F5_B_INPUTS and [o.name for o in dit.get_outputs()] == list(_F5_B_ALIAS)
    a_in = [a.name for a in pre.get_inputs()]
    a_out = dict(zip([o.name for o in pre.get_outputs()],
                     pre.run(None, {a_in[0]: g["pre_audio"].reshape(1, 1, -1), a_in[1]: g["pre_text_ids"].reshape(1, -1),
                                    a_in[2]: np.array([int(g["pre_N"])], np.int64)})))
    host = {k: a_out[k] for k in _F5_B_INPUTS[1:7]}
    host["noise"], host["time_step"] = g["dit_noise"][None].copy(), np.array([0], np.int32)       # the fixture's noise from here on
    on_dev = lambda x: onnxruntime.OrtValue.ortvalue_from_numpy(x, "cuda", 0)
    dev = {k: on_dev(host[k].copy() if k in _F5_B_ALIAS.values() else host[k]) for k in _F5_B_INPUTS}
    for k, v in dev.items():
        assert v.is_device() and v.device_name() == "cuda" and v.shape() == list(host[k].shape), k
    cos_q = host["rope_cos_q"]
    assert dev["rope_cos_q"]._t.untyped_storage().nbytes() == cos_q.shape[2] * cos_q.shape[3] * 4     # the broadcast axes did not cross PCIe
    binding = dit.io_binding()
    for k in _F5_B_INPUTS:
        binding.bind_ortvalue_input(name=k, ortvalue=dev[k])
    for out_name, in_name in _F5_B_ALIAS.items():
        binding.bind_ortvalue_output(name=out_name, ortvalue=dev[in_name])
    steps = cfg.nfe_step - 1
    where = dev["noise"].data_ptr()
    for _ in range(steps):
        dit.run_with_iobinding(binding)
    bound = binding.get_outputs()
    assert bound[0] is dev["noise"] and dev["noise"].data_ptr() == where                      # advanced in place, as bound
    assert int(bound[1].numpy()[0]) == steps and int(dev["time_step"]._t.cpu()[0]) == steps   # the device value itself, not just a mirror
    latent_dev = bound[0].numpy()
    # the same steps on host arrays through run(): same engine -> the same bits
    x, t = host["noise"].copy(), host["time_step"].copy()
    for _ in range(steps):
        x, t = dit.run(list(_F5_B_ALIAS), {**host, "noise": x, "time_step": t})
    assert np.array_equal(latent_dev, x)
    c_in = [a.name for a in voc.get_inputs()]
    wave = voc.run(None, {c_in[0]: latent_dev, c_in[1]: a_out["ref_signal_len"]})[0]
    err = np.sqrt(np.mean(((wave[0, 0].astype(np.float64) - g["e2e_i16"]) / 32767.0) ** 2))
    assert err < 5e-4, err
    # unbound outputs: inputs untouched, outputs are NEW device values
    x0, t0 = on_dev(g["dit_noise"][None].copy()), on_dev(np.array([0], np.int32))
    feed = {**{k: dev[k] for k in _F5_B_INPUTS[1:7]}, "noise": x0, "time_step": t0}
    x1, t1 = dit.run_with_ort_values(list(_F5_B_ALIAS), feed)
    assert x1.is_device() and x1 is not x0 and np.array_equal(x0.numpy(), g["dit_noise"][None])
    assert int(t1.numpy()[0]) == 1 and int(t0.numpy()[0]) == 0
    one_step, _ = dit.run(list(_F5_B_ALIAS), {**host, "noise": g["dit_noise"][None].copy(), "time_step": np.array([0], np.int32)})
    assert np.array_equal(x1.numpy(), one_step)
    # a RoPE value that is not graph A's table is refused on the device path too; a wrong element type as well
    for key, value in (("rope_cos_q", on_dev(np.ascontiguousarray(cos_q) * 0.5)), ("noise", onnxruntime.OrtValue._from_tensor(x0._t.double()))):
        with pytest.raises(onnxruntime.InvalidArgument):
            dit.run_with_ort_values(None, {**feed, key: value})


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
