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


def _segment(text: str) -> List[str]:
    if all(ord(c) < 128 for c in text):
        # jieba.cut on pure-ASCII text yields runs of [A-Za-z0-9] (its re_eng buffer) and every other
        # character (punctuation, each whitespace char) as a single-character token
        return re.findall(r"[a-zA-Z0-9]+|.", text, flags=re.S)
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
