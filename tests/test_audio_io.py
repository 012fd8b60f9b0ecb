"""Host-side audio I/O (mi355tts/audio_io.py) pinned against CPython's own audioop — the module pydub calls for
``set_channels(1)`` / ``set_frame_rate`` in the reference's prompt loader (F5-TTS-ONNX-Inference.py:223)."""
import struct
import wave

import numpy as np
import pytest

from mi355tts import audio_io as A

audioop = pytest.importorskip("audioop")


@pytest.mark.parametrize("inrate,outrate,n", [(44100, 24000, 5000), (16000, 24000, 3001), (48000, 24000, 4096),
                                              (22050, 24000, 777), (24000, 24000, 100), (8000, 24000, 1), (44100, 24000, 2)])
def test_ratecv_bit_exact(inrate, outrate, n):
    rng = np.random.default_rng(inrate + n)
    x = rng.integers(-32768, 32768, size=n, dtype=np.int64).astype(np.int16)
    ref, _ = audioop.ratecv(x.tobytes(), 2, 1, inrate, outrate, None)
    got = A.ratecv(x, inrate, outrate)
    np.testing.assert_array_equal(got, np.frombuffer(ref, dtype=np.int16))


def test_tomono_bit_exact():
    rng = np.random.default_rng(3)
    x = rng.integers(-32768, 32768, size=(4001, 2), dtype=np.int64).astype(np.int16)
    ref = audioop.tomono(x.tobytes(), 2, 0.5, 0.5)
    np.testing.assert_array_equal(A.tomono(x), np.frombuffer(ref, dtype=np.int16))


def test_load_prompt_like_the_reference(tmp_path):
    rng = np.random.default_rng(5)
    x = rng.integers(-20000, 20000, size=(22050, 2), dtype=np.int64).astype(np.int16)
    p = str(tmp_path / "stereo_44k.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(x.tobytes())
    mono = audioop.tomono(x.tobytes(), 2, 0.5, 0.5)
    ref, _ = audioop.ratecv(mono, 2, 1, 44100, 24000, None)
    got = A.load_prompt(p, 24000)
    assert got.dtype == np.int16 and got.shape[:2] == (1, 1)
    np.testing.assert_array_equal(got.reshape(-1), np.frombuffer(ref, dtype=np.int16))


def test_wavex_writer_round_trip(tmp_path):
    rng = np.random.default_rng(7)
    y = rng.integers(-32768, 32768, size=(1, 1, 12345), dtype=np.int64).astype(np.int16)
    p = str(tmp_path / "out.wav")
    A.write_wavex(p, y.reshape(-1), 24000)
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    size, tag, ch, rate, brate, align, bits, cb, valid, mask = struct.unpack("<IHHIIHHHHI", raw[16:44])
    assert (size, tag, ch, rate, brate, align, bits, cb, valid, mask) == (40, 0xFFFE, 1, 24000, 48000, 2, 16, 22, 16, 4)
    assert raw[44:60] == bytes.fromhex("0100000000001000800000aa00389b71")
    assert raw[60:64] == b"fact" and struct.unpack("<II", raw[64:72]) == (4, 12345)
    back, r = A.read_wav(p)
    assert r == 24000
    np.testing.assert_array_equal(back.reshape(-1), y.reshape(-1))
    A.write_wavex(p, y, 24000)                       # the engine's (1, 1, L) output shape is accepted as is
    np.testing.assert_array_equal(A.read_wav(p)[0].reshape(-1), y.reshape(-1))
    with pytest.raises(ValueError):
        A.write_wavex(p, y.astype(np.float32), 24000)


def test_read_wav_other_widths(tmp_path):
    p = str(tmp_path / "u8.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(1); w.setframerate(8000); w.writeframes(bytes([0, 128, 255]))
    x, r = A.read_wav(p)
    assert r == 8000 and x.reshape(-1).tolist() == [-32768, 0, 127 << 8]
    p = str(tmp_path / "s24.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(3); w.setframerate(8000)
        w.writeframes(b"".join(int(v).to_bytes(3, "little", signed=True) for v in (-8388608, 256, 8388607)))
    assert A.read_wav(p)[0].reshape(-1).tolist() == [-32768, 1, 32767]
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.wav"), "wb").write(b"nope")
        A.read_wav(str(tmp_path / "bad.wav"))
