#!/usr/bin/env python
"""Golden fixture for the IndexTTS text front end: runs the REFERENCE's own functions (exec'ed from
IndexTTS/Inference_IndexTTS_ONNX.py:95-577 where they lie) on a set of inputs with

  * identity stand-ins for the un-vendored WeTextProcessing normalisers (so the fixture pins everything AROUND them:
    placeholders, punctuation map, language choice), and
  * a tiny sentencepiece BPE model trained here on a seeded corpus (tests/golden/indextts_sp.model, a data file).

Output: tests/golden/indextts_text.json (inputs + the reference's outputs).  Build container only."""
import io
import json
import os
import platform
import re
import sys
import traceback
import warnings
from functools import lru_cache
from typing import List, Union, overload

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R                      # noqa: E402

CORPUS = [
    "HELLO WORLD , THIS IS A TEST OF THE TOKENIZER .", "THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG !",
    "你 好 世 界 , 今 天 天 气 很 好 .", "大 家 好 , 我 是 语 音 合 成 系 统 . 请 问 你 是 谁 ?",
    "SEE YOU LATER - MAYBE TOMORROW ' OK '", "一 二 三 四 五 六 七 八 九 十 ,", "XUAN4 ZE2 QV4 JV2 …",
    "ARE YOU SURE ? YES , I AM SURE . REALLY ! …", "中 文 和 ENGLISH 混 合 的 句 子 .",
] * 8

TEXTS = [
    "你好世界是 hello world 的中文", "Hello, world! How are you today? I'm fine; thanks.", "今天天气很好：我们去公园吧！好不好？",
    "这是一个很长的句子，它有很多逗号，所以可以在逗号处分割，如果超过了限制，就会被分成几段，然后再合并。", "xuan4 ze2 is pinyin, so is qu4 and ju2.",
    "克里斯托弗·诺兰导演了这部电影", "a@b.com", "《书名》（括号）【方括号】「引号」", "wait... really,,, ok……", "嗯，呣", "A", " ", "",
    "No punctuation at all here just words going on and on", "one - two - three - four", "He said 'go.' Then left.",
    "\u201ccurly\u201d and \u2018single\u2019 quotes (parens) [brackets]: colon; semicolon",
]


class Identity:
    def normalize(self, s):
        return s


def main():
    model_prefix = os.path.join(HERE, "indextts_sp")
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS), model_writer=buf, vocab_size=140, model_type="bpe",
                                   character_coverage=1.0, bos_id=0, eos_id=1, unk_id=2, pad_id=-1,
                                   normalization_rule_name="identity")
    with open(model_prefix + ".model", "wb") as f:
        f.write(buf.getvalue())
    ns = {"re": re, "os": os, "platform": platform, "traceback": traceback, "warnings": warnings, "List": List, "Union": Union,
          "overload": overload, "lru_cache": lru_cache, "SentencePieceProcessor": spm.SentencePieceProcessor}
    R.exec_lines(R.REF + "/IndexTTS/Inference_IndexTTS_ONNX.py", 95, 577, ns)
    norm = ns["TextNormalizer"]()
    norm.zh_normalizer, norm.en_normalizer = Identity(), Identity()
    norm.load = lambda: None
    tok = ns["TextTokenizer"](model_prefix + ".model", norm)
    out = {"texts": TEXTS, "cjk": [], "cjk_lower": [], "use_chinese": [], "normalized": [], "ids": [], "pieces": [], "decoded": [],
           "split_40": [], "split_120": []}
    for t in TEXTS:
        out["cjk"].append(ns["tokenize_by_CJK_char"](t))
        out["cjk_lower"].append(ns["tokenize_by_CJK_char"](t, do_upper_case=False))
        out["use_chinese"].append(bool(norm.use_chinese(t)))
        out["normalized"].append(norm.normalize(t))
        ids = tok.encode(t)
        pieces = tok.tokenize(t)
        out["ids"].append([int(i) for i in ids])
        out["pieces"].append(pieces)
        out["decoded"].append(tok.decode([int(i) for i in ids]) if ids else "")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out["split_40"].append(tok.split_sentences(pieces, 40))
            out["split_120"].append(tok.split_sentences(pieces, 120))
    # the splitter on hand-made token lists (quote continuation, lone marks, oversize without commas)
    cases = [["▁A", "B", ".", "'", "C", "!"], [".", "▁", ".", "X", "?"], ["W"] * 30 + ["."], ["A", ",", "B", ",", "C", "-", "D", "."] * 5,
             ["A"] * 8 + ["-"] + ["B"] * 8 + ["."]]
    # (a hyphen- or comma-delimited piece that is STILL longer than the cap recurses forever in the reference, :440-447 ->
    #  :426; not a fixture case)
    out["split_cases"] = cases
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["split_cases_out"] = [ns["TextTokenizer"].split_sentences_by_token(list(c), tok.punctuation_marks_tokens, 10) for c in cases]
    out["detok"] = [["你 好 世 界 是 HELLO WORLD 的 中 文", False], ["SEE YOU!", True], ["A-B C 你 好", True]]
    out["detok_out"] = [ns["de_tokenized_by_CJK_char"](a, do_lower_case=b) for a, b in out["detok"]]
    out["pinyin"] = ["xuan4", "ju2", "QU4", "lve4", "xun1", "jue2"]
    out["pinyin_out"] = [norm.correct_pinyin(p) for p in out["pinyin"]]
    with open(os.path.join(HERE, "indextts_text.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("indextts_text.json:", len(TEXTS), "texts; sp vocab", tok.vocab_size)
    for t, n, p in list(zip(TEXTS, out["normalized"], out["pieces"]))[:6]:
        print(repr(t), "->", repr(n), p[:12])


if __name__ == "__main__":
    main()
