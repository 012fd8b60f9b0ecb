"""Host-side text front end of the IndexTTS driver (the caller side of graph B).

Restates /root/reference IndexTTS/Inference_IndexTTS_ONNX.py:
  tokenize_by_CJK_char / de_tokenized_by_CJK_char   :95-171   (CJK characters become single tokens, the rest is upper-cased)
  TextNormalizer                                    :174-353  (pinyin-with-tone and hyphenated-name placeholders around the
                                                               third-party normaliser, then the punctuation map)
  TextTokenizer                                     :356-577  (sentencepiece ids, sentence split at . ! ? with the
                                                               120-token cap, comma / hyphen fallback, merge of short pieces)
The zh / en normalisers themselves are WeTextProcessing (`tn`) / `wetext` — un-vendored, unpinned and not installable
here: they are injected (``TextNormalizer(zh=..., en=...)``); ``load()`` imports them like the reference does and raises
if they are absent.  Token ids must equal the reference's for the same sentencepiece model, so the rules below follow the
reference's observable behaviour exactly (tests/golden/make_golden_text.py runs the reference functions for the fixture).
"""
from __future__ import annotations

import re
import warnings
from typing import Callable, List, Optional, Sequence

# one CJK character per token: Hangul Jamo, CJK radicals .. Yi, Hangul syllables, compatibility ideographs / forms,
# half-width katakana & hangul, supplementary ideographic plane
_CJK = re.compile("([ᄀ-ᇿ⺀-꓏ꡀ-힯豈-﫿︰-﹏･-ￜ\U00020000-\U0002FFFF])")
_EN_RUN = re.compile(r"([A-Z]+(?:[\s-][A-Z-]+)*)", re.IGNORECASE)
_SENT_TAG = re.compile(r"^.*?(<sent_(\d+)>)")


def tokenize_by_CJK_char(line: str, do_upper_case: bool = True) -> str:
    line = line.strip()
    if not line:
        return ""
    parts = [w.strip() for w in _CJK.split(line)]
    parts = [w for w in parts if w]
    return " ".join(w.upper() for w in parts) if do_upper_case else " ".join(parts)


def de_tokenized_by_CJK_char(line: str, do_lower_case: bool = False) -> str:
    runs = _EN_RUN.findall(line)
    if not runs:
        return "".join(line.split())
    tmp = line
    for i, run in enumerate(runs):                       # protect latin runs (they contain spaces) from the re-join
        tmp = tmp.replace(run, f"<sent_{i}>", 1)
    words = tmp.split()
    for j, w in enumerate(words):
        m = _SENT_TAG.match(w)
        if m:
            w = w.replace(m.group(1), runs[int(m.group(2))])
            words[j] = w.lower() if do_lower_case else w
    return "".join(words)


# punctuation map applied after normalisation (order matters: it becomes one alternation, first match wins at a position).
# The reference's table spells its two curly-double-quote entries as `(""", "'"), (""", "'")`, which Python reads as ONE
# entry whose key is the text between the triple quotes, and its two curly-single-quote entries are plain apostrophes
# (:186): typographic quotes are therefore NOT mapped — reproduced (the odd key is kept for the alternation order).
_CHAR_REP = [
    ("：", ","), ("；", ","), (";", ","), ("，", ","), ("。", "."), ("！", "!"), ("？", "?"), ("\n", " "), ("·", "-"), ("、", ","),
    ("...", "…"), (",,,", "…"), ("，，，", "…"), ("……", "…"), (", \"'\"), (", "'"), ('"', "'"), ("'", "'"),
    ("（", "'"), ("）", "'"), ("(", "'"), (")", "'"), ("《", "'"), ("》", "'"), ("【", "'"), ("】", "'"), ("[", "'"), ("]", "'"),
    ("—", "-"), ("～", "-"), ("~", "-"), ("「", "'"), ("」", "'"), (":", ","),
]


class TextNormalizer:
    """``zh`` / ``en``: objects with ``.normalize(str) -> str`` (WeTextProcessing / wetext normalisers)."""

    def __init__(self, zh=None, en=None):
        self.zh_normalizer, self.en_normalizer = zh, en
        self.char_rep_map = dict(_CHAR_REP)
        self.zh_char_rep_map = {"$": ".", **self.char_rep_map}
        self._email = re.compile(r"^[a-zA-Z0-9]+@[a-zA-Z0-9]+\.[a-zA-Z]+$")
        self._han = re.compile("[一-鿿]")
        self._alpha = re.compile(r"[a-zA-Z]")
        self._pinyin = re.compile(r"([bmnpqdfghjklzcsxwy]?h?[aeiouüv]{1,2}[ng]*|ng)([1-5])", re.IGNORECASE)
        self._name = re.compile("[一-鿿]+([-·—][一-鿿]+){1,2}")
        self._jqx = re.compile(r"([jqx])[uü](n|e|an)*(\d)", re.IGNORECASE)
        self._rep = re.compile("|".join(re.escape(k) for k in self.char_rep_map))
        self._zh_rep = re.compile("|".join(re.escape(k) for k in self.zh_char_rep_map))

    def load(self):
        """Import the third-party normalisers the way the reference does (:233-242)."""
        import platform
        if platform.system() == "Darwin":
            from wetext import Normalizer
            self.zh_normalizer = Normalizer(remove_erhua=False, lang="zh", operator="tn")
            self.en_normalizer = Normalizer(lang="en", operator="tn")
        else:
            from tn.chinese.normalizer import Normalizer as Zh
            from tn.english.normalizer import Normalizer as En
            self.zh_normalizer = Zh(remove_interjections=False, remove_erhua=False, overwrite_cache=False)
            self.en_normalizer = En(overwrite_cache=False)

    def match_email(self, s: str) -> bool:
        return self._email.match(s) is not None

    def use_chinese(self, s: str) -> bool:
        if self._han.search(s) or not self._alpha.search(s) or self.match_email(s):
            return True
        return self._pinyin.search(s) is not None            # latin text with tone-numbered pinyin goes the zh way too

    # placeholders: the normaliser must not touch tone-numbered pinyin ("xuan4") and hyphenated names
    @staticmethod
    def _protect(text: str, found: Sequence[str], tag: str):
        uniq = list(dict.fromkeys(found))
        if not uniq:
            return text, None
        for i, item in enumerate(uniq):
            text = text.replace(item, f"<{tag}_{chr(ord('a') + i)}>")
        return text, uniq

    def save_pinyin_tones(self, text: str):
        return self._protect(text, ["".join(p) for p in self._pinyin.findall(text)], "pinyin")

    def save_names(self, text: str):
        # findall returns the LAST repetition of the group only ("-二" of "一-二"): the reference protects exactly that piece
        return self._protect(text, ["".join(n) for n in self._name.findall(text)], "n")

    def correct_pinyin(self, pinyin: str) -> str:
        if pinyin[0].lower() not in "jqx":
            return pinyin
        return self._jqx.sub(r"\g<1>v\g<2>\g<3>", pinyin).upper()       # ju / qu / xu -> jv / qv / xv (ü), upper case

    def restore_names(self, text: str, names) -> str:
        for i, name in enumerate(names or ()):
            text = text.replace(f"<n_{chr(ord('a') + i)}>", name)
        return text

    def restore_pinyin_tones(self, text: str, pinyins) -> str:
        for i, p in enumerate(pinyins or ()):
            text = text.replace(f"<pinyin_{chr(ord('a') + i)}>", self.correct_pinyin(p))
        return text

    def normalize(self, text: str) -> str:
        text = text.replace("嗯", "恩").replace("呣", "母")
        if not self.zh_normalizer or not self.en_normalizer:
            raise RuntimeError("TextNormalizer: no zh / en normaliser (call load(), or pass zh= / en= objects)")
        text = text.rstrip()
        if self.use_chinese(text):
            t, pinyins = self.save_pinyin_tones(text)
            t, names = self.save_names(t)
            try:
                out = self.zh_normalizer.normalize(t)
            except Exception:                                # the reference prints the traceback and goes on with ""
                out = ""
            out = self.restore_pinyin_tones(self.restore_names(out, names), pinyins)
            return self._zh_rep.sub(lambda m: self.zh_char_rep_map[m.group()], out)
        try:
            out = self.en_normalizer.normalize(text)
        except Exception:
            out = text
        return self._rep.sub(lambda m: self.char_rep_map[m.group()], out)


class TextTokenizer:
    """sentencepiece ids the way the reference produces them; ``sp`` is a loaded ``SentencePieceProcessor`` or a model path."""
    punctuation_marks_tokens = [".", "!", "?", "▁.", "▁?", "▁..."]
    bos_token, eos_token, unk_token, pad_token = "<s>", "</s>", "<unk>", None
    bos_token_id, eos_token_id, pad_token_id = 0, 1, -1

    def __init__(self, sp, normalizer: Optional[TextNormalizer] = None):
        if isinstance(sp, str):
            from sentencepiece import SentencePieceProcessor
            sp = SentencePieceProcessor(model_file=sp)
        self.sp_model, self.normalizer = sp, normalizer
        self.pre_tokenizers: List[Callable[[str], str]] = [tokenize_by_CJK_char]

    @property
    def vocab_size(self) -> int:
        return self.sp_model.GetPieceSize()

    @property
    def unk_token_id(self) -> int:
        return self.sp_model.unk_id()

    def convert_ids_to_tokens(self, ids):
        return self.sp_model.IdToPiece(ids)

    def convert_tokens_to_ids(self, tokens):
        return [self.sp_model.PieceToId(t) for t in ([tokens] if isinstance(tokens, str) else tokens)]

    def encode(self, text: str, out_type=int):
        if not text:
            return []
        if len(text.strip()) == 1:                           # single characters skip normalisation and the CJK split
            return self.sp_model.Encode(text, out_type=out_type)
        if self.normalizer:
            text = self.normalizer.normalize(text)
        for pre in self.pre_tokenizers:
            text = pre(text)
        return self.sp_model.Encode(text, out_type=out_type)

    def tokenize(self, text: str) -> List[str]:
        return self.encode(text, out_type=str)

    def decode(self, ids, do_lower_case: bool = False) -> str:
        return de_tokenized_by_CJK_char(self.sp_model.Decode([ids] if isinstance(ids, int) else ids), do_lower_case)

    # ---- sentence split: the GPT prompt of one sentence is capped at max_tokens_per_sentence (120) text tokens -------------
    @staticmethod
    def split_sentences_by_token(tokens: Sequence[str], split_tokens: Sequence[str], max_tokens: int) -> List[List[str]]:
        if not tokens:
            return []
        marks = set(split_tokens)
        out: List[List[str]] = []
        cur: List[str] = []
        i, n = 0, len(tokens)
        while i < n:
            tok = tokens[i]
            cur.append(tok)
            if tok in marks:
                if len(cur) <= 1 or (len(cur) == 2 and cur[0] == "▁"):      # a lone mark: dropped
                    cur = []
                    i += 1
                    continue
                if i + 1 < n and tokens[i + 1] in ("'", "▁'"):              # closing quote stays with its sentence
                    cur.append(tokens[i + 1])
                    i += 1
                if len(cur) <= max_tokens:
                    out.append(cur)
                else:
                    out.extend(TextTokenizer._split_long(cur, max_tokens))
                cur = []
            i += 1
        if cur:
            out.append(cur)
        return TextTokenizer._merge_short(out, max_tokens)

    @staticmethod
    def _split_long(sentence: List[str], max_tokens: int) -> List[List[str]]:
        if "," in sentence or "▁," in sentence:
            return TextTokenizer.split_sentences_by_token(sentence, [",", "▁,"], max_tokens)
        if "-" in sentence:
            return TextTokenizer.split_sentences_by_token(sentence, ["-"], max_tokens)
        warnings.warn(f"sentence of {len(sentence)} tokens exceeds the limit of {max_tokens} and has no comma / hyphen to split at",
                      RuntimeWarning)
        return [sentence[:max_tokens], sentence[max_tokens:]]

    @staticmethod
    def _merge_short(sentences: List[List[str]], max_tokens: int) -> List[List[str]]:
        merged: List[List[str]] = []
        for s in sentences:
            if not s:
                continue
            if merged and len(merged[-1]) + len(s) <= max_tokens:
                merged[-1].extend(s)
            else:
                merged.append(s)
        return merged

    def split_sentences(self, tokenized: Sequence[str], max_tokens_per_sentence: int = 120) -> List[List[str]]:
        return self.split_sentences_by_token(tokenized, self.punctuation_marks_tokens, max_tokens_per_sentence)
