"""CPU: host-side logic — text front end, ORT-compat metadata, sharding helpers."""
import json
import os

import numpy as np
import pytest

from mi355tts import text as T
from mi355tts import shard as S
from mi355tts import weights as W
from mi355tts import ort_compat as ORT
from mi355tts.config import BigVGANConfig, F5Config


def test_list_str_to_idx_matches_reference_function(golden_dir):
    g = np.load(os.path.join(golden_dir, "f5_small.npz"))
    vocab = W.synth_vocab(2545)
    cases = [list("ab c!"), list("Z"), ["a", "é", "zhong1", " ", "b"]]
    ids = T.list_str_to_idx(cases, vocab)
    assert ids.dtype == np.int32
    assert np.array_equal(ids, g["g10_ids"])            # OOV -> 0, ragged rows padded with -1


def test_convert_char_to_pinyin_ascii_branch():
    out = T.convert_char_to_pinyin(["Some call me nature; others call me."])
    assert "".join(out[0]) == "Some call me nature, others call me."        # ';' -> ','
    # the reference inserts a space before a multi-char ASCII segment that directly follows punctuation
    assert "".join(T.convert_char_to_pinyin(["a,bc"])[0]) == "a, bc"
    assert "".join(T.convert_char_to_pinyin(["it's"])[0]) == "it's"          # ... but not after ' : " or space
    assert T.convert_char_to_pinyin([""]) == [[]]


def test_max_duration_formula():
    # 6.0 s reference audio, equal-length texts -> N = 2 * 563 (BASELINE config 3)
    assert T.max_duration(144000, "a" * 77, "b" * 77) == 1126
    assert T.max_duration(144000, "a" * 77, "b" * 77, speed=2.0) == 563 + 281
    # the punctuation bonus uses a literal (non-character-class) pattern: it never fires on normal text
    assert T.max_duration(1000, "你好。", "你好。") == T.max_duration(1000, "你好x", "你好x")


def test_vocab_loader(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_text(" \na\nb\n", encoding="utf-8")
    assert T.load_vocab(str(p)) == {" ": 0, "a": 1, "b": 2}


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [S.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.shard_range(4, 2, 2)


def test_bucket_by_length():
    b = S.bucket_by_length([10, 12, 10, 10, 12], max_batch=2)
    assert sorted(map(tuple, b)) == [(0, 2), (1, 4), (3,)]


def test_graph_io_names_match_the_exports():
    f5 = F5Config()
    i, o = ORT._graph_io("F5_Preprocess", f5, "f32")
    assert [a.name for a in i] == ["audio", "text_ids", "max_duration"]
    assert [a.name for a in o] == ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text",
                                   "cat_mel_text_drop", "ref_signal_len"]
    assert o[1].shape == [2, 16, "max_duration", 64] and o[3].shape == [2, 16, 64, "max_duration"]
    assert o[5].shape == [1, "max_duration", 612]
    i, o = ORT._graph_io("F5_Transformer", f5, "bf16")
    assert [a.name for a in i][-1] == "time_step" and [a.name for a in o] == ["denoised", "time_step"]
    i, o = ORT._graph_io("F5_Decode", f5, "f32")
    assert [a.name for a in i] == ["denoised", "ref_signal_len"] and o[0].type == "tensor(int16)"
    i, o = ORT._graph_io("BigVGAN", BigVGANConfig(), "f16")
    assert i[0].name == "mel_features" and i[0].shape[1] == 100 and o[0].name == "generated_wav"


def test_session_options_and_errors(tmp_path):
    so = ORT.SessionOptions()
    so.intra_op_num_threads = 8
    so.graph_optimization_level = ORT.GraphOptimizationLevel.ORT_ENABLE_ALL
    so.add_session_config_entry("session.set_denormal_as_zero", "1")
    assert so.config_entries["session.set_denormal_as_zero"] == "1"
    with pytest.raises(ORT.InvalidArgument):
        so.add_session_config_entry("k", 1)
    with pytest.raises(ORT.Fail):
        ORT.InferenceSession(str(tmp_path / "missing.onnx"))
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"format": "x"}))
    with pytest.raises(ORT.InvalidArgument):
        ORT.InferenceSession(str(bad))
    ORT.set_seed(9527)
    v = ORT.OrtValue.ortvalue_from_numpy(np.ones((1, 2)), "cpu", 0)
    assert v.numpy().shape == (1, 2)


def test_manifest_roundtrip(tmp_path):
    cfg = BigVGANConfig.small()
    w = tmp_path / "w.npy"
    np.save(w, np.zeros(4, np.float32))
    m = ORT.save_model(str(tmp_path / "BigVGAN.mi355.json"), "BigVGAN", cfg, str(w), "f16")
    man = json.load(open(m))
    assert man["graph"] == "BigVGAN" and man["weights"] == "w.npy" and man["config"]["num_mels"] == cfg.num_mels


def test_f5_fold_matches_spec_counts():
    cfg = F5Config.small()
    raw = W.synth_state(W.f5_spec(cfg), 1)
    blob = W.pack_f5(cfg, raw)
    assert blob.size == sum(int(np.prod(s)) for _, s, _ in W.f5_packed_spec(cfg))
    st = W.fold_f5(cfg, raw)
    sf = cfg.dim_head ** -0.25
    k = "transformer.transformer_blocks.0.attn.to_q.weight"
    np.testing.assert_allclose(st[k], raw[k] * np.float32(sf), rtol=1e-6)
    k2 = "vocos.backbone.convnext.0.pwconv2.weight"
    np.testing.assert_allclose(st[k2], raw["vocos.backbone.convnext.0.gamma"][:, None] * raw[k2], rtol=1e-6)
    assert not any(k.endswith(".gamma") and k.startswith("vocos") for k in st)


def test_indextts_graph_f_io_names_match_the_export():
    cfg = BigVGANConfig.indextts()
    i, o = ORT._graph_io("IndexTTS_F", cfg, "f16")
    assert [a.name for a in i] == [f"save_bigvgan_conds_{k}" for k in range(6)] + \
        ["bigvgan_cond_layer_speaker_embedding", "save_hidden_state"]
    assert [a.shape[1] for a in i[:6]] == [768, 384, 192, 96, 48, 24] and i[6].shape == [1, 1536, 1]
    assert i[7].shape == ["kv_seq_len", 1280] and o[0].name == "generated_wav"
    assert cfg.hop == 1024 and cfg.out_len(126) == 126 * 1024 + 30
    n_spec = sum(int(np.prod(s)) for _, s, _ in W.bigvgan_spec(cfg))
    assert n_spec == sum(int(np.prod(s)) for _, s, _ in W.bigvgan_spec(BigVGANConfig(
        num_mels=1280, upsample_rates=cfg.upsample_rates, upsample_kernel_sizes=cfg.upsample_kernel_sizes,
        use_bias_at_final=True))) + 2 * 1280


def test_ascii_segmentation_follows_jieba_block_rules():
    """jieba.cut on pure-ASCII text (un-vendored; restated in mi355tts.text): blocks of [a-zA-Z0-9+#&._%-] go through the DAG
    + finalseg's `[a-zA-Z0-9]+(?:\\.\\d+)?%?` split, so decimals, percentages and punctuation runs are single multi-character
    tokens — and convert_char_to_pinyin (F5-TTS-ONNX-Inference.py:116-119) puts a space in front of those."""
    from mi355tts import text as T
    seg = T._segment
    assert seg("wait... ok") == ["wait", "...", " ", "ok"]
    assert seg("pi is 3.14") == ["pi", " ", "is", " ", "3.14"]
    assert seg("C++ and c#") == ["C++", " ", "and", " ", "c#"]
    assert seg("a--b") == ["a", "--", "b"]
    assert seg("50% off, AT&T.") == ["50%", " ", "off", ",", " ", "AT&T", "."]
    assert seg("I'm") == ["I", "'", "m"]
    assert seg("a\r\nb") == ["a", "\r\n", "b"]
    j = lambda t: "".join(T.convert_char_to_pinyin([t])[0])
    assert j("wait... ok") == "wait ... ok"
    assert j("pi is 3.14") == "pi is 3.14"
    assert j("a... b") == "a ... b"
    assert j("C++") == "C++" and j("use C++") == "use C++"
    assert j("x;y") == "x,y"
    assert j("Hello, world!") == "Hello, world!"
