"""CPU: the numpy oracle (oracle/bigvgan_np.py) against the golden vectors produced by the
reference's own module code (tests/golden/make_golden.py -> bigvgan_small.npz)."""
import os

import numpy as np
import pytest

from mi355tts.config import BigVGANConfig
from mi355tts import weights as W
from oracle import bigvgan_np as O


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "bigvgan_small.npz"))


@pytest.fixture(scope="module")
def small():
    cfg = BigVGANConfig.small()
    return cfg, W.synth_state(W.bigvgan_spec(cfg), 9527)


def test_filter_taps(g):
    h = O.aa_filter()
    assert h.shape == (12,)
    np.testing.assert_allclose(h, g["taps_up"], atol=1e-7)
    np.testing.assert_allclose(h, g["taps_down"], atol=1e-7)
    assert abs(float(h.sum()) - 1.0) < 1e-6


def test_activation1d_block_and_post(g):
    h = O.aa_filter()
    y = O.activation1d(g["act_x"], g["act_alpha"], g["act_beta"], h)
    np.testing.assert_allclose(y, g["act_y"], atol=5e-6)
    yp = O.activation1d(g["post_x"], g["post_alpha"], g["post_beta"], h, post=True)
    assert yp.shape[-1] == g["post_x"].shape[-1] + 30         # the +30 quirk
    np.testing.assert_allclose(yp, g["post_y"], atol=5e-6)


def test_ampblock1(g, small):
    cfg, st = small
    h = O.aa_filter()
    for j, k in enumerate(cfg.resblock_kernel_sizes):
        y = O.amp_block1(g["amp_x"], st, j, k, cfg.resblock_dilation_sizes[j], h)
        np.testing.assert_allclose(y, g[f"amp_y{j}"], atol=2e-5)


@pytest.mark.parametrize("name", ["a", "b"])
def test_generator_and_int16(g, small, name):
    cfg, st = small
    mel = g[f"gen_mel_{name}"]
    y = O.generator(cfg, st, mel)
    assert y.shape == (mel.shape[0], 1, cfg.out_len(mel.shape[2]))
    np.testing.assert_allclose(y, g[f"gen_y_{name}"], atol=2e-6)
    w = O.bigvgan_int16(cfg, st, mel)
    assert w.dtype == np.int16
    # float rounding can flip a truncation boundary: allow 1 LSB
    assert np.abs(w.astype(np.int32) - g[f"gen_i16_{name}"].astype(np.int32)).max() <= 1


def test_reference_smoke_input_ones(g, small):
    cfg, st = small
    w = O.bigvgan_int16(cfg, st, g["gen_mel_ones"])
    assert np.abs(w.astype(np.int32) - g["gen_i16_ones"].astype(np.int32)).max() <= 1


def test_weight_norm_removal_roundtrip():
    rng = np.random.default_rng(0)
    v = rng.standard_normal((4, 3, 5)).astype(np.float32)
    gg = rng.standard_normal((4, 1, 1)).astype(np.float32)
    out = W.remove_weight_norm_state({"c.weight_g": gg, "c.weight_v": v, "c.bias": np.zeros(4, np.float32)})
    nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
    np.testing.assert_allclose(out["c.weight"], v * gg / nrm, rtol=1e-6)
    assert "c.bias" in out and "c.weight_v" not in out


def test_synth_weights_deterministic():
    a = W.synth_normal(9527, "x", (1000,))
    b = W.synth_normal(9527, "x", (1000,))
    assert np.array_equal(a, b)
    assert abs(float(a.mean())) < 0.15 and 0.85 < float(a.std()) < 1.15
    # pinned values: the generator is counter-based and must never change
    np.testing.assert_allclose(W.synth_normal(1, "pin", (3,)), W.synth_normal(1, "pin", (5,))[:3])


def test_indextts_graph_f_against_reference_wrapper(golden_dir):
    """IndexTTS_F.forward (Export_IndexTTS.py:292-314) exec'ed from the reference + its modeling_modified/models.py:
    full-size speaker-conditioned vocoder (k = u transposed convs, biased conv_post, pre-LayerNorm)."""
    g = np.load(os.path.join(golden_dir, "indextts_f.npz"))
    cfg = BigVGANConfig.indextts()
    st = W.synth_state(W.bigvgan_spec(cfg), 9527)
    conds = [g[f"cond{i}"].reshape(-1) for i in range(cfg.num_upsamples)] + [g["cond_pre"].reshape(-1)]
    w = O.indextts_f_int16(cfg, st, g["latent"], conds)
    assert w.shape == g["wav_i16"].shape == (1, 1, 3 * 1024 + 30)
    assert np.abs(w.astype(np.int32) - g["wav_i16"].astype(np.int32)).max() <= 2
    assert np.sqrt(np.mean(g["wav_i16"].astype(np.float64) ** 2)) > 1000


def test_oracle_full_architecture_against_reference_fixture(golden_dir):
    """The FULL BigVGAN-v2 architecture (112 M parameters, 6 stages) through the oracle against the reference generator +
    int16 wrapper run in the build container (tests/golden/make_golden_full.py), on the reference's own smoke input
    np.ones (BigVGAN/Export_BigVGAN.py:165) at 64 frames — what pins the oracle at real widths."""
    import os
    from mi355tts.config import BigVGANConfig
    g = np.load(os.path.join(golden_dir, "bigvgan_full.npz"))
    cfg = BigVGANConfig()
    st = W.synth_state(W.bigvgan_spec(cfg), 9527)
    w = O.bigvgan_int16(cfg, st, np.ones((1, cfg.num_mels, 64), np.float32))
    assert w.shape == (1, 1, 64 * 256 + 30)
    assert np.abs(w[0, 0].astype(np.int32) - g["ones64_i16"].astype(np.int32)).max() <= 2
