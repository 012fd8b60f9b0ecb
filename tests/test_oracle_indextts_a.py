"""CPU: the numpy oracle of IndexTTS graph A against the fixture the REFERENCE wrapper produced
(tests/golden/make_golden_indextts_a.py: IndexTTS/Export_IndexTTS.py:60-200 exec'd over stand-in sub-modules)."""
import os

import numpy as np
import pytest

from mi355tts.config import IndexCondConfig
from mi355tts import weights as W
from oracle import indextts_a_np as OA


@pytest.fixture(scope="module")
def ga(golden_dir):
    return np.load(os.path.join(golden_dir, "indextts_a.npz"))


@pytest.mark.parametrize("tag", ["s_", "r_"])
def test_graph_a_oracle_matches_reference_wrapper(ga, tag):
    cfg = IndexCondConfig.small()
    st = W.fold_cond(cfg, W.synth_state(W.cond_spec(cfg), 9527))
    conds, cond0, latent, mel = OA.graph_a(cfg, st, ga[tag + "audio"])
    assert mel.shape == (cfg.n_mels, cfg.frames(len(ga[tag + "audio"])))
    assert latent.shape == (cfg.latents, cfg.model_dim)
    np.testing.assert_allclose(latent, ga[tag + "conds_latent"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(cond0, ga[tag + "cond_layer"], atol=1e-4, rtol=1e-4)
    for i, c in enumerate(conds):
        np.testing.assert_allclose(c, ga[tag + f"cond_{i}"], atol=1e-4, rtol=1e-4)
    # the outputs are not degenerate: the conditioning actually depends on the audio
    assert np.abs(ga["s_conds_latent"] - ga["r_conds_latent"]).max() > 1e-2 and np.abs(ga["s_cond_layer"]).max() > 1e-2


def test_fold_changes_only_what_the_wrapper_scales():
    cfg = IndexCondConfig.small()
    raw = W.synth_state(W.cond_spec(cfg), 9527)
    st = W.fold_cond(cfg, raw)
    changed = {k for k in raw if not np.array_equal(raw[k], st[k])}
    for k in changed:
        assert ("embed.out.0" in k or "self_attn.linear_q" in k or "self_attn.linear_k" in k or "linear_pos" in k or "pos_bias_" in k
                or "to_q.weight" in k or "to_kv.weight" in k), k
    assert W.pack_cond(cfg, raw).size == sum(int(np.prod(s)) for _, s, _ in W.cond_spec(cfg))
