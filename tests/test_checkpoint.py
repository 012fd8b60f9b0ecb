"""CPU: checkpoint readers (mi355tts/checkpoint.py) — files written here in the upstream key layouts from the seeded state
must pack to the same blob as the plain dict (Export_F5.py:207-221, bigvgan.py:505-514,
vocos/pretrained.py:62-79)."""
import numpy as np
import pytest
import torch

from mi355tts.config import BigVGANConfig, F5Config
from mi355tts import weights as W
from mi355tts import checkpoint as CK


def test_safetensors_reader_matches_the_reference_library(tmp_path):
    st = {"a.weight": W.synth_normal(1, "a", (5, 7)), "b": np.arange(6, dtype=np.int64).reshape(2, 3),
          "h": W.synth_normal(2, "h", (4,)).astype(np.float16)}
    CK.write_safetensors(str(tmp_path / "t.safetensors"), st, {"format": "pt"})
    back = CK.read_safetensors(str(tmp_path / "t.safetensors"))
    assert set(back) == set(st) and all(np.array_equal(back[k], st[k]) and back[k].dtype == st[k].dtype for k in st)
    safetensors = pytest.importorskip("safetensors.torch")
    lib = safetensors.load_file(str(tmp_path / "t.safetensors"))                      # our writer is readable by the library
    assert all(np.array_equal(lib[k].numpy(), st[k]) for k in st)
    bf = torch.from_numpy(st["a.weight"]).to(torch.bfloat16)
    safetensors.save_file({"x": bf, "y": torch.from_numpy(st["a.weight"])}, str(tmp_path / "lib.safetensors"))
    got = CK.read_safetensors(str(tmp_path / "lib.safetensors"))                      # and the library's files by our reader
    assert np.array_equal(got["x"], bf.float().numpy()) and np.array_equal(got["y"], st["a.weight"])


def test_f5_and_vocos_checkpoints_in_upstream_layout(tmp_path):
    cfg = F5Config.small()
    state = W.synth_state(W.f5_spec(cfg), 9527)
    want = W.pack_f5(cfg, state)
    # model_*.safetensors: EMA copy, `ema_model.` prefix, EMA bookkeeping + mel-spectrogram buffers the loader drops
    ema = {"ema_model." + k: v for k, v in state.items() if k.startswith("transformer.")}
    ema["ema_model.initted"] = np.array([1.0], np.float32)
    ema["ema_model.step"] = np.array([1250000], np.int64)
    ema["ema_model.mel_spec.mel_stft.mel_scale.fb"] = np.zeros((513, 100), np.float32)
    ema["ema_model.mel_spec.mel_stft.spectrogram.window"] = np.zeros((1024,), np.float32)
    CK.write_safetensors(str(tmp_path / "model_1250000.safetensors"), ema)
    # Vocos pytorch_model.bin: plain torch state dict with the feature extractor's buffers
    voc = {k[len("vocos."):]: torch.from_numpy(v) for k, v in state.items() if k.startswith("vocos.")}
    voc["feature_extractor.mel_spec.spectrogram.window"] = torch.zeros(1024)
    voc["feature_extractor.mel_spec.mel_scale.fb"] = torch.zeros(513, 100)
    torch.save(voc, tmp_path / "pytorch_model.bin")
    got = CK.pack_f5_from_files(cfg, str(tmp_path / "model_1250000.safetensors"), str(tmp_path / "pytorch_model.bin"))
    assert np.array_equal(got, want)
    # the .pt flavour of the same checkpoint: nested under ema_model_state_dict
    torch.save({"ema_model_state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in ema.items()}, "step": 1250000},
               tmp_path / "model_1250000.pt")
    assert np.array_equal(CK.pack_f5_from_files(cfg, str(tmp_path / "model_1250000.pt"), str(tmp_path / "pytorch_model.bin")), want)
    # a checkpoint of another width is refused, not silently mis-packed
    with pytest.raises((ValueError, KeyError)):
        CK.pack_f5_from_files(F5Config(dim=256, heads=4, depth=2, text_dim=64, text_num_embeds=40, conv_layers=1, pos_conv_groups=4,
                                       vocos_dim=64, vocos_intermediate=128, vocos_layers=1),
                              str(tmp_path / "model_1250000.safetensors"), str(tmp_path / "pytorch_model.bin"))
    with pytest.raises(ValueError):
        CK.f5_transformer_state({"foo": np.zeros(3)})


@pytest.mark.parametrize("flavour", ["plain", "weight_g_v", "parametrizations"])
def test_bigvgan_generator_checkpoint(tmp_path, flavour):
    cfg = BigVGANConfig.small()
    state = W.synth_state(W.bigvgan_spec(cfg), 9527)
    want = W.pack_bigvgan(cfg, state)
    sd = {}
    for k, v in state.items():
        t = torch.from_numpy(v)
        is_conv_w = k.endswith(".weight") and v.ndim == 3
        if flavour == "plain" or not is_conv_w:
            sd[k] = t
        else:
            # weight = g * v / ||v|| (dim 0): choose v = 1.7 * weight, g = ||weight||
            nrm = t.flatten(1).norm(dim=1).view(-1, 1, 1)
            names = (".weight_g", ".weight_v") if flavour == "weight_g_v" else (".parametrizations.weight.original0", ".parametrizations.weight.original1")
            sd[k[: -len(".weight")] + names[0]] = nrm
            sd[k[: -len(".weight")] + names[1]] = 1.7 * t
    sd["resblocks.0.activations.0.upsample.filter"] = torch.zeros(1, 1, 12)        # registered FIR buffers ride along
    torch.save({"generator": sd}, tmp_path / "bigvgan_generator.pt")
    got = CK.pack_bigvgan_from_file(cfg, str(tmp_path / "bigvgan_generator.pt"))
    if flavour == "plain":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
    torch.save(sd, tmp_path / "bare.pt")                                             # a bare state dict is accepted too
    np.testing.assert_allclose(CK.pack_bigvgan_from_file(cfg, str(tmp_path / "bare.pt")), want, rtol=2e-6, atol=1e-7)


def test_safetensors_reader_rejects_malformed_files(tmp_path):
    """the header is untrusted input — oversize header length, offsets outside the file, size / shape mismatch,
    truncation and unsupported dtypes all raise a ValueError naming the problem (no huge allocation, no opaque reshape)."""
    import json
    import struct
    from mi355tts import checkpoint as C

    def write(name, header, data=b"", hlen=None):
        h = json.dumps(header).encode()
        p = tmp_path / name
        p.write_bytes(struct.pack("<Q", len(h) if hlen is None else hlen) + h + data)
        return str(p)

    good = {"a": {"dtype": "F32", "shape": [2, 2], "data_offsets": [0, 16]}}
    assert C.read_safetensors(write("ok.safetensors", good, b"\0" * 16))["a"].shape == (2, 2)
    with pytest.raises(ValueError, match="header length"):
        C.read_safetensors(write("hlen.safetensors", good, b"\0" * 16, hlen=1 << 40))
    with pytest.raises(ValueError, match="tensor a.*outside"):
        C.read_safetensors(write("trunc.safetensors", good, b"\0" * 8))
    bad = {"a": {"dtype": "F32", "shape": [3, 2], "data_offsets": [0, 16]}}
    with pytest.raises(ValueError, match="tensor a.*needs 24"):
        C.read_safetensors(write("shape.safetensors", bad, b"\0" * 16))
    bad = {"ok": good["a"], "z": {"dtype": "F8_E4M3", "shape": [4], "data_offsets": [16, 20]}}
    with pytest.raises(ValueError, match="tensor z.*unsupported dtype"):
        C.read_safetensors(write("dtype.safetensors", bad, b"\0" * 20))
    bad = {"a": {"dtype": "F32", "shape": [2, 2], "data_offsets": [8, 4]}}
    with pytest.raises(ValueError, match="tensor a"):
        C.read_safetensors(write("order.safetensors", bad, b"\0" * 16))
    (tmp_path / "short.safetensors").write_bytes(b"\x01\x02")
    with pytest.raises(ValueError, match="shorter"):
        C.read_safetensors(str(tmp_path / "short.safetensors"))
