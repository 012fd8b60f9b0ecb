"""Host-side text front end of the F5 driver (the caller side of graph A).

Restates /root/reference F5_TTS/F5-TTS-ONNX-Inference.py:
  vocab.txt loader            :88-92    (line i, minus its newline, maps to index i)
  convert_char_to_pinyin      :96-136   (jieba segmentation + pypinyin for CJK; pure-ASCII text needs neither)
  list_str_to_idx             :140-148  (OOV -> 0, pad -1)
  duration heuristic          :227-231  (the zh-punctuation bonus uses a literal pattern and is therefore
                                         always 0 for normal text — reproduced, not "fixed")
jieba / pypinyin are optional: they are imported lazily and only when the text contains non-ASCII
characters; without them CJK input raises (no silent fallback).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Sequence

import numpy as np

ZH_PAUSE_PUNC = r"。，、；：？！"


def load_vocab(path: str) -> Dict[str, int]:
    vocab = {}
    with open(path, "r", encoding="utf-8") as f:
        for i, line in enumerate(f):
            vocab[line[:-1]] = i
    return vocab


# ---- jieba.cut(text) (cut_all=False, HMM=True) restated for pure-ASCII text -------------------------------------------
# jieba (un-vendored, unpinned; behaviour of 0.42.1) splits the sentence into blocks with
#   re_han_default = ([\u4E00-\u9FD5a-zA-Z0-9+#&\._%\-]+)
# and, between blocks, re_skip_default = (\r\n|\s): whitespace tokens are yielded whole, everything else char by char.
# A block goes through __cut_DAG: the max-probability route over dictionary words (dict.txt's only pure-ASCII entries are
# AT&T, c#, C#, c++, C++, frequency 3 of 60101967), single-character routes are buffered and a buffer of two or more
# characters that is not itself a dictionary word is handed to finalseg.cut, whose non-CJK branch splits on
#   re_skip = ([a-zA-Z0-9]+(?:\.\d+)?%?)
# keeping both the matches and the runs between them ("3.14", "50%", "..." and "--" are single tokens).
_RE_BLOCK = re.compile(r"([a-zA-Z0-9+#&\._%\-]+)")
_RE_SKIP_DEFAULT = re.compile(r"(\r\n|\s)")
_RE_FINALSEG_SKIP = re.compile(r"([a-zA-Z0-9]+(?:\.\d+)?%?)")
_ASCII_DICT = {"AT&T": 3, "c#": 3, "C#": 3, "c++": 3, "C++": 3}
_LOG_TOTAL = math.log(60101967)


def _cut_dag_ascii(blk: str) -> List[str]:
    n = len(blk)
    route = [(0.0, 0)] * (n + 1)
    for i in range(n - 1, -1, -1):                       # Tokenizer.calc: best (log-probability, end index) from i
        best = (-_LOG_TOTAL + route[i + 1][0], i)        # log(FREQ.get(char) or 1) = 0
        for w, f in _ASCII_DICT.items():
            if blk.startswith(w, i):
                cand = (math.log(f) - _LOG_TOTAL + route[i + len(w)][0], i + len(w) - 1)
                if cand > best:
                    best = cand
        route[i] = best
    out: List[str] = []
    buf = ""

    def flush():
        nonlocal buf
        if buf:
            if len(buf) == 1:
                out.append(buf)
            else:
                out.extend(x for x in _RE_FINALSEG_SKIP.split(buf) if x)
            buf = ""
    x = 0
    while x < n:
        y = route[x][1] + 1
        if y - x == 1:
            buf += blk[x:y]
        else:
            flush()
            out.append(blk[x:y])
        x = y
    flush()
    return out


def _segment_ascii(text: str) -> List[str]:
    out: List[str] = []
    for blk in _RE_BLOCK.split(text):
        if not blk:
            continue
        if _RE_BLOCK.match(blk):
            out.extend(_cut_dag_ascii(blk))
        else:
            for x in _RE_SKIP_DEFAULT.split(blk):
                if _RE_SKIP_DEFAULT.match(x):
                    out.append(x)
                else:
                    out.extend(x)
    return out


def _segment(text: str) -> List[str]:
    if all(ord(c) < 128 for c in text):
        return _segment_ascii(text)
    try:
        import jieba
    except ImportError as e:                                 # pragma: no cover
        raise RuntimeError("non-ASCII text needs `jieba` (and `pypinyin`) like the reference driver") from e
    if jieba.dt.initialized is False:
        jieba.default_logger.setLevel(50)
        jieba.initialize()
    return list(jieba.cut(text))


def convert_char_to_pinyin(text_list: Sequence[str], polyphone: bool = True) -> List[List[str]]:
    final = []
    trans = str.maketrans({";": ",", "“": '"', "”": '"', "‘": "'", "’": "'"})

    def is_chinese(c):
        return "㄀" <= c <= "鿿"

    for text in text_list:
        chars: List[str] = []
        text = text.translate(trans)
        for seg in _segment(text):
            nbytes = len(seg.encode("utf-8"))
            if nbytes == len(seg):                            # pure alphabets and symbols
                if chars and nbytes > 1 and chars[-1] not in " :'\"":
                    chars.append(" ")
                chars.extend(seg)
            else:
                from pypinyin import lazy_pinyin, Style       # CJK path
                if polyphone and nbytes == 3 * len(seg):
                    py = lazy_pinyin(seg, style=Style.TONE3, tone_sandhi=True)
                    for i, c in enumerate(seg):
                        if is_chinese(c):
                            chars.append(" ")
                        chars.append(py[i])
                else:
                    for c in seg:
                        if ord(c) < 256:
                            chars.extend(c)
                        elif is_chinese(c):
                            chars.append(" ")
                            chars.extend(lazy_pinyin(c, style=Style.TONE3, tone_sandhi=True))
                        else:
                            chars.append(c)
        final.append(chars)
    return final


def list_str_to_idx(text: Sequence[Sequence[str]], vocab: Dict[str, int], padding_value: int = -1) -> np.ndarray:
    rows = [[vocab.get(c, 0) for c in t] for t in text]
    width = max((len(r) for r in rows), default=0)
    out = np.full((len(rows), width), padding_value, dtype=np.int32)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def max_duration(audio_len: int, ref_text: str, gen_text: str, hop_length: int = 256, speed: float = 1.0) -> int:
    ref_len = len(ref_text.encode("utf-8")) + 3 * len(re.findall(ZH_PAUSE_PUNC, ref_text))
    gen_len = len(gen_text.encode("utf-8")) + 3 * len(re.findall(ZH_PAUSE_PUNC, gen_text))
    ref_audio_len = audio_len // hop_length + 1
    return ref_audio_len + int(ref_audio_len / ref_len * gen_len / speed)
