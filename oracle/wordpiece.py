"""Pure-Python restatement of the BERT (uncased) WordPiece tokenizer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference sends raw text as ``"prompt"`` (main.py:139-142); tokenisation happens
inside Ollama/llama.cpp (absent from /root/reference, unpinned) -> PARITY UNPINNED by
the reference.  This restates the published BERT algorithm (clean -> CJK spacing ->
lower-case -> NFD + strip Mn -> whitespace/punctuation split -> greedy longest-match
WordPiece, ``[CLS]`` ... ``[SEP]``, truncation) and is cross-checked against
``tokenizers.BertWordPieceTokenizer`` on a synthetic local vocab in
tests/test_oracle_wordpiece.py.
"""
from __future__ import annotations

import unicodedata
from typing import Dict, List


def is_whitespace(ch: str) -> bool:
    if ch in (" ", "\t", "\n", "\r"):
        return True
    return unicodedata.category(ch) == "Zs"


def is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch).startswith("C")


def is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF
            or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF
            or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_tokens(text: str) -> List[str]:
    cleaned = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or is_control(ch):
            continue
        if is_whitespace(ch):
            cleaned.append(" ")
        elif is_cjk(cp):
            cleaned += [" ", ch, " "]
        else:
            cleaned.append(ch)
    # strip accents first (NFD, drop Mn) then lower-case: the order tokenizers'
    # BertNormalizer applies them
    stripped = "".join(c for c in unicodedata.normalize("NFD", "".join(cleaned))
                       if unicodedata.category(c) != "Mn")
    lowered = stripped.lower()
    toks: List[str] = []
    for word in lowered.split():
        cur = ""
        for ch in word:
            if is_punctuation(ch):
                if cur:
                    toks.append(cur)
                    cur = ""
                toks.append(ch)
            else:
                cur += ch
        if cur:
            toks.append(cur)
    return toks


def wordpiece(word: str, vocab: Dict[str, int], unk: str = "[UNK]", max_chars: int = 100) -> List[int]:
    if len(word) > max_chars:
        return [vocab[unk]]
    out: List[int] = []
    start = 0
    while start < len(word):
        end = len(word)
        cur = None
        while start < end:
            sub = word[start:end]
            if start > 0:
                sub = "##" + sub
            if sub in vocab:
                cur = vocab[sub]
                break
            end -= 1
        if cur is None:
            return [vocab[unk]]
        out.append(cur)
        start = end
    return out


def encode(text: str, vocab: Dict[str, int], max_len: int = 512) -> List[int]:
    """-> [CLS] pieces... [SEP], truncated to ``max_len`` ids including both specials."""
    ids: List[int] = []
    for w in basic_tokens(text):
        ids += wordpiece(w, vocab)
    ids = ids[: max(max_len - 2, 0)]
    return [vocab["[CLS]"]] + ids + [vocab["[SEP]"]]


def synthetic_vocab(texts: List[str], size: int = 3000) -> List[str]:
    """Deterministic local vocab (no fetch): specials, single characters (and their
    ## forms), then the most frequent word prefixes / suffix pieces of ``texts``."""
    from collections import Counter
    specials = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars: Counter = Counter()
    pieces: Counter = Counter()
    for t in texts:
        for w in basic_tokens(t):
            chars.update(w)
            for n in (2, 3, 4, 5, 7):
                if len(w) >= n:
                    pieces[w[:n]] += 1
                    pieces["##" + w[-n:]] += 1
                if len(w) > n + 1:
                    pieces["##" + w[1:1 + n]] += 1
            pieces[w] += 2
    vocab = list(specials)
    seen = set(vocab)
    for c, _ in sorted(chars.items(), key=lambda kv: (-kv[1], kv[0]))[:400]:
        for tok in (c, "##" + c):
            if tok not in seen:
                vocab.append(tok)
                seen.add(tok)
    for p, _ in sorted(pieces.items(), key=lambda kv: (-kv[1], kv[0])):
        if len(vocab) >= size:
            break
        if p not in seen:
            vocab.append(p)
            seen.add(p)
    return vocab
