"""GPU: IndexTTS graph A (cond.hip through the C-ABI) against the REFERENCE wrapper's fixture (small model) and against the
numpy oracle at the published IndexTTS-1.5 widths (Conformer 6 x 512 / Perceiver 32 x 1280 / ECAPA 512..1536)."""
import os

import numpy as np
import pytest

from mi355tts.config import IndexCondConfig
from mi355tts import weights as W
from mi355tts.indextts import IndexCond
from oracle import indextts_a_np as OA

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["s_", "r_"])
def test_graph_a_against_reference_fixture(golden_dir, tag):
    ga = np.load(os.path.join(golden_dir, "indextts_a.npz"))
    cfg = IndexCondConfig.small()
    raw = W.synth_state(W.cond_spec(cfg), 9527)
    eng = IndexCond(cfg, raw)
    conds, lat, mel = eng.run(ga[tag + "audio"], return_mel=True)
    st = W.fold_cond(cfg, raw)
    ref_mel = OA.mel_front_end(cfg, st, ga[tag + "audio"])
    m = ref_mel > np.log(2e-5)               # log of a clamped value: compared where the reference is above the clamp floor
    assert m.mean() > 0.5
    np.testing.assert_allclose(mel.T[m], ref_mel[m], atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(lat, ga[tag + "conds_latent"], atol=5e-4, rtol=1e-3)
    outs, cond0 = eng.split_conds(conds)
    np.testing.assert_allclose(cond0.reshape(-1), ga[tag + "cond_layer"], atol=2e-4, rtol=1e-3)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.reshape(-1), ga[tag + f"cond_{i}"], atol=2e-4, rtol=1e-3)
    c2, l2 = eng.run(ga[tag + "audio"].reshape(1, 1, -1))
    assert np.array_equal(c2, conds) and np.array_equal(l2, lat)                                          # run-to-run identity
    eng.close()


def test_graph_a_full_width_against_oracle():
    """IndexTTS-1.5 widths, 2 s of prompt audio (frames 197, encoder length 98): every product on the MFMA launcher."""
    cfg = IndexCondConfig()
    raw = W.synth_state(W.cond_spec(cfg), 9527, fast=True)
    st = W.fold_cond(cfg, raw)
    rng = np.random.default_rng(5)
    t = np.arange(48000) / 24000.0
    audio = np.clip(0.25 * 32767 * np.sin(2 * np.pi * 140.0 * t) * (1 + 0.4 * np.sin(2 * np.pi * 2.0 * t)) + rng.normal(0, 800, t.size), -32768, 32767).astype(np.int16)
    conds_o, cond0_o, lat_o, _ = OA.graph_a(cfg, st, audio)
    eng = IndexCond(cfg, raw)
    conds, lat = eng.run(audio)
    assert lat.shape == (32, 1280) and conds.shape == (1536 + 768 + 384 + 192 + 96 + 48 + 24,)
    assert rel(lat, lat_o) < 2e-3, rel(lat, lat_o)
    ref = np.concatenate([cond0_o] + list(conds_o))
    assert rel(conds, ref) < 2e-3, rel(conds, ref)
    assert np.abs(lat_o).max() > 0.1 and np.abs(ref).max() > 0.01
    with pytest.raises(Exception):
        eng.run(np.zeros(10, np.int16))                                                                  # too short for the k15 / k5 stacks
    eng.close()
