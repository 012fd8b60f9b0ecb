"""CPU: the IndexTTS text front end (mi355tts/indextts_text.py) against outputs of the reference's own functions
(tests/golden/make_golden_text.py exec's IndexTTS/Inference_IndexTTS_ONNX.py:95-577 where it lies; identity stand-ins for the
un-vendored WeTextProcessing normalisers; a tiny seeded sentencepiece model, tests/golden/indextts_sp.model)."""
import json
import os
import warnings

import pytest

from mi355tts import indextts_text as T


class Identity:
    def normalize(self, s):
        return s


@pytest.fixture(scope="module")
def g(golden_dir):
    with open(os.path.join(golden_dir, "indextts_text.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok(golden_dir):
    return T.TextTokenizer(os.path.join(golden_dir, "indextts_sp.model"), T.TextNormalizer(zh=Identity(), en=Identity()))


def test_cjk_pre_tokenizer_and_inverse(g):
    for t, up, lo in zip(g["texts"], g["cjk"], g["cjk_lower"]):
        assert T.tokenize_by_CJK_char(t) == up
        assert T.tokenize_by_CJK_char(t, do_upper_case=False) == lo
    for (line, lower), want in zip(g["detok"], g["detok_out"]):
        assert T.de_tokenized_by_CJK_char(line, do_lower_case=lower) == want


def test_normalizer_placeholders_language_choice_and_punctuation_map(g, tok):
    n = tok.normalizer
    for t, uc, want in zip(g["texts"], g["use_chinese"], g["normalized"]):
        assert n.use_chinese(t) == uc, t
        assert n.normalize(t) == want, t
    assert [n.correct_pinyin(p) for p in g["pinyin"]] == g["pinyin_out"]
    with pytest.raises(RuntimeError):
        T.TextNormalizer().normalize("no backends loaded")


def test_token_ids_pieces_and_decode(g, tok):
    for t, ids, pieces, dec in zip(g["texts"], g["ids"], g["pieces"], g["decoded"]):
        assert [int(i) for i in tok.encode(t)] == ids, t
        assert tok.tokenize(t) == pieces
        if ids:
            assert tok.decode(ids) == dec
    assert tok.bos_token_id == 0 and tok.eos_token_id == 1 and tok.vocab_size == 140


def test_sentence_split_cap_quote_continuation_and_merge(g, tok):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for pieces, s40, s120 in zip(g["pieces"], g["split_40"], g["split_120"]):
            assert tok.split_sentences(pieces, 40) == s40
            assert tok.split_sentences(pieces, 120) == s120
            assert all(len(s) <= 120 for s in s120)
        for case, want in zip(g["split_cases"], g["split_cases_out"]):
            assert T.TextTokenizer.split_sentences_by_token(list(case), tok.punctuation_marks_tokens, 10) == want
